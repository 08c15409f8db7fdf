// Host-layer exerciser built with -fsanitize=address,undefined (see `make sanitize` in ray_tracing_amd/csrc):
// walks the C ABI's CPU entry points over the bundled assets, malformed inputs included.  Exit code 0 = no error found.
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <string.h>
#include <math.h>

#include <string>
#include <vector>

#include "../include/ray_tracing_amd.h"
#include "../ray_tracing_amd/csrc/rt_host.h"  // prepare_scene: the host half of rt_scene_create (the device half is not linked here)

#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) {                                                      \
            fprintf(stderr, "CHECK failed %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #cond, rt_last_error()); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

int main(int argc, char** argv) {
    const std::string assets = argc > 1 ? argv[1] : "assets";
    const std::string tmp = argc > 2 ? argv[2] : "/tmp";
    rt_material m;
    rt_material_default(&m);
    m.baseColor[0] = 1, m.baseColor[1] = 0.73f, m.baseColor[2] = 0.25f, m.roughness = 0.5f, m.metallic = 1;
    float rot[3] = {0, 0, 0}, tr[3] = {0.3f, -1.6f, 0}, sc[3] = {1.5f, 1.5f, 1.5f}, T[16];
    CHECK(rt_transform_matrix(rot, tr, sc, T) == RT_OK);

    rt_mesh* mesh = nullptr;
    CHECK(rt_mesh_create(&mesh) == RT_OK);
    CHECK(rt_mesh_read_obj(mesh, (assets + "/models/quad.obj").c_str(), &m, T, 0) == RT_OK);
    CHECK(rt_mesh_read_obj(mesh, (assets + "/models/sphere2.obj").c_str(), &m, T, 1) == RT_OK);
    CHECK(rt_mesh_read_obj(mesh, (assets + "/models/does_not_exist.obj").c_str(), &m, T, 1) == RT_ERR_IO);
    CHECK(rt_mesh_read_obj_chap2(mesh, (assets + "/models_chap2/quad.obj").c_str(), 5.0f, 0.5f) == RT_OK);
    const int nT = rt_mesh_count(mesh);
    CHECK(nT > 20000);
    for (int sah = 0; sah < 2; ++sah)
        for (int leaf : {1, 8, 100000}) {
            float* nodes = nullptr;
            int nN = 0;
            CHECK(rt_bvh_build(mesh, leaf, sah, &nodes, &nN) == RT_OK);
            CHECK(nN >= 2 && rt_mesh_count(mesh) == nT);
            if (leaf == 8 && sah == 1) {
                const std::string p = tmp + "/rt_sanitize.rtsc";
                const uint64_t key = rt_hash_fnv1a64("k", 1, 0);
                CHECK(rt_scene_cache_save(p.c_str(), key, rt_mesh_data(mesh), nT, nodes, nN) == RT_OK);
                float *t2 = nullptr, *n2 = nullptr;
                int a = 0, b = 0;
                CHECK(rt_scene_cache_load(p.c_str(), key, &t2, &a, &n2, &b) == RT_OK && a == nT && b == nN);
                CHECK(memcmp(t2, rt_mesh_data(mesh), (size_t)nT * 144) == 0 && memcmp(n2, nodes, (size_t)nN * 48) == 0);
                rt_free(t2), rt_free(n2);
                CHECK(rt_scene_cache_load(p.c_str(), key + 1, &t2, &a, &n2, &b) == RT_ERR_FORMAT);
                FILE* f = fopen(p.c_str(), "r+b");  // truncate: header only
                CHECK(f != nullptr);
                fclose(f);
                CHECK(truncate(p.c_str(), 60) == 0);
                CHECK(rt_scene_cache_load(p.c_str(), key, &t2, &a, &n2, &b) == RT_ERR_FORMAT);
                remove(p.c_str());
                // the host half of rt_scene_create: records, margins, entry BVH (forced) with its climb tables
                for (int entry_mode : {0, 1, -1}) {
                    rta::ScenePrep prep;
                    CHECK(rta::prepare_scene(rt_mesh_data(mesh), nT, nodes, nN, nullptr, nullptr, 0, 0, entry_mode, &prep) == RT_OK);
                    CHECK(prep.has_root && prep.ttri.size() == (size_t)nT * 12 && prep.has_entry == (entry_mode == 1));
                    CHECK(!prep.has_entry || (prep.leafx.size() == (size_t)nT && prep.map2.size() == (size_t)nT));
                }
                // malformed trees must be refused (or rendered un-culled), never walked out of bounds
                std::vector<float> bad(nodes, nodes + (size_t)nN * 12);
                rta::ScenePrep prep;
                bad[12 + 0] = (float)(nN + 5);  // root's left child out of range
                CHECK(rta::prepare_scene(rt_mesh_data(mesh), nT, bad.data(), nN, nullptr, nullptr, 0, 0, 1, &prep) == RT_ERR_BAD_BVH);
                bad[12 + 0] = 1.0f;             // the root is its own child
                CHECK(rta::prepare_scene(rt_mesh_data(mesh), nT, bad.data(), nN, nullptr, nullptr, 0, 0, 1, &prep) == RT_ERR_BAD_BVH);
                bad[12 + 0] = 2.5f;             // not an index
                CHECK(rta::prepare_scene(rt_mesh_data(mesh), nT, bad.data(), nN, nullptr, nullptr, 0, 0, 1, &prep) == RT_ERR_BAD_BVH);
                bad.assign(nodes, nodes + (size_t)nN * 12);
                for (int k = 1; k < nN; ++k)
                    if (bad[(size_t)k * 12 + 3] > 0.0f) {  // first leaf: its range runs past the triangle array
                        bad[(size_t)k * 12 + 4] = (float)(nT - 1);
                        bad[(size_t)k * 12 + 3] = 8.0f;
                        break;
                    }
                CHECK(rta::prepare_scene(rt_mesh_data(mesh), nT, bad.data(), nN, nullptr, nullptr, 0, 0, 1, &prep) == RT_ERR_BAD_BVH);
                bad.assign(nodes, nodes + (size_t)nN * 12);
                bad[2 * 12 + 6] = NAN, bad[3 * 12 + 9] = -INFINITY;  // non-finite boxes: accepted, rendered without culling or entry BVH
                CHECK(rta::prepare_scene(rt_mesh_data(mesh), nT, bad.data(), nN, nullptr, nullptr, 0, 0, 1, &prep) == RT_OK);
                CHECK(!prep.has_entry && !(prep.cull_eps < INFINITY));
            }
            rt_free(nodes);
        }
    // a mesh large enough for the builders' all-threads path (exact_sort.h, chunked sweeps): 2 x 300 x 240 lattice triangles,
    // centroid coordinates tie by the thousand
    {
        const std::string p = tmp + "/rt_sanitize_lattice.obj";
        FILE* f = fopen(p.c_str(), "w");
        CHECK(f != nullptr);
        const int nx = 300, nz = 240;
        for (int i = 0; i <= nx; ++i)
            for (int j = 0; j <= nz; ++j) fprintf(f, "v %d %d %d\n", i, (i * 7 + j * 3) % 5, j);
        for (int i = 0; i < nx; ++i)
            for (int j = 0; j < nz; ++j) {
                const int a = i * (nz + 1) + j + 1, b = a + nz + 1;
                fprintf(f, "f %d %d %d\nf %d %d %d\n", a, b, b + 1, a, b + 1, a + 1);
            }
        fclose(f);
        rt_mesh* big = nullptr;
        CHECK(rt_mesh_create(&big) == RT_OK);
        float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        CHECK(rt_mesh_read_obj(big, p.c_str(), &m, I, 0) == RT_OK && rt_mesh_count(big) == 2 * nx * nz);
        remove(p.c_str());
        for (int sah = 0; sah < 2; ++sah) {
            float* bn = nullptr;
            int bnN = 0;
            CHECK(rt_bvh_build(big, 8, sah, &bn, &bnN) == RT_OK && bnN > 2 * nx * nz / 8);
            if (sah) {
                rta::ScenePrep prep;
                CHECK(rta::prepare_scene(rt_mesh_data(big), rt_mesh_count(big), bn, bnN, nullptr, nullptr, 0, 0, 1, &prep) == RT_OK);
                CHECK(prep.has_entry && prep.rec2.size() % 32 == 0 && prep.stack2 > 0);
            }
            rt_free(bn);
        }
        rt_mesh_destroy(big);
    }
    // empty mesh
    rt_mesh* empty = nullptr;
    CHECK(rt_mesh_create(&empty) == RT_OK);
    float* nodes = nullptr;
    int nN = 0;
    CHECK(rt_bvh_build(empty, 8, 1, &nodes, &nN) == RT_OK && nN == 1);
    rt_free(nodes);
    rt_mesh_destroy(empty);
    rt_mesh_destroy(mesh);

    // malformed OBJ files
    {
        const std::string p = tmp + "/rt_sanitize_bad.obj";
        FILE* f = fopen(p.c_str(), "w");
        CHECK(f != nullptr);
        fputs("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n", f);  // face index past the vertex list
        fclose(f);
        rt_mesh* b = nullptr;
        CHECK(rt_mesh_create(&b) == RT_OK);
        float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        CHECK(rt_mesh_read_obj(b, p.c_str(), &m, I, 1) == RT_ERR_FORMAT);
        f = fopen(p.c_str(), "w");
        fputs("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1/1/1 2 3\n", f);
        fclose(f);
        CHECK(rt_mesh_read_obj(b, p.c_str(), &m, I, 1) != RT_OK);
        rt_mesh_destroy(b);
        remove(p.c_str());
    }
    // HDR: load, cache, malformed header
    int W = 0, H = 0;
    float* hdr = nullptr;
    CHECK(rt_hdr_load((assets + "/HDR/peppermint_powerplant_4k.hdr").c_str(), &W, &H, &hdr) == RT_OK && W == 1024 && H == 512);
    std::vector<float> cache((size_t)W * H * 3);
    CHECK(rt_hdr_cache(hdr, W, H, cache.data()) == RT_OK);
    {  // HDR texels + SampleHdr directions of the host preparation, on a scene without triangles
        rta::ScenePrep prep;
        CHECK(rta::prepare_scene(nullptr, 0, nullptr, 0, hdr, cache.data(), W, H, 0, &prep) == RT_OK);
        CHECK(!prep.has_root && prep.hdr4.size() == (size_t)W * H * 4 && prep.cdir4.size() == (size_t)W * H * 4);
    }
    std::vector<unsigned char> rgb8((size_t)W * H * 3);
    CHECK(rt_tonemap_rgb8(hdr, W, H, rgb8.data(), 1) == RT_OK);
    CHECK(rt_write_png((tmp + "/rt_sanitize.png").c_str(), W, H, rgb8.data(), 0) == RT_OK);
    remove((tmp + "/rt_sanitize.png").c_str());
    rt_free(hdr);
    {
        const std::string p = tmp + "/rt_sanitize_bad.hdr";
        FILE* f = fopen(p.c_str(), "wb");
        const char a[] = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 4 +X 4\n\x01\x01\x01\xff\x01\x01\x01\xff";  // old RLE: repeats before any pixel
        fwrite(a, 1, sizeof(a) - 1, f);
        fclose(f);
        // like the reference (hdr:87-88) a short or garbled pixel stream leaves the remaining rows black; it must stay in bounds
        CHECK(rt_hdr_load(p.c_str(), &W, &H, &hdr) == RT_OK && W == 4 && H == 4);
        rt_free(hdr);
        f = fopen(p.c_str(), "wb");
        const char b[] = "#?RADIANCE\nX\n\n-Y 2 +X 16\n\x02\x02\x00\x10\xff\x07\xff\x07\xff\x07\xff\x07";  // new RLE, runs of 127 into 16 pixels
        fwrite(b, 1, sizeof(b) - 1, f);
        fclose(f);
        CHECK(rt_hdr_load(p.c_str(), &W, &H, &hdr) == RT_OK && W == 16 && H == 2);
        rt_free(hdr);
        f = fopen(p.c_str(), "wb");
        // old RLE: one pixel, then six consecutive (1,1,1,n) repeat markers, some with a zero count: the shift of the
        // repeat count must stay defined (UBSan) and the decode in bounds
        const char c[] = "#?RADIANCE\nX\n\n-Y 2 +X 4\n\x10\x20\x30\x80\x01\x01\x01\x00\x01\x01\x01\x00\x01\x01\x01\x00\x01\x01\x01\x00\x01\x01\x01\xff\x01\x01\x01\xff";
        fwrite(c, 1, sizeof(c) - 1, f);
        fclose(f);
        CHECK(rt_hdr_load(p.c_str(), &W, &H, &hdr) == RT_OK && W == 4 && H == 2);
        rt_free(hdr);
        f = fopen(p.c_str(), "wb");
        fputs("not an hdr file", f);
        fclose(f);
        CHECK(rt_hdr_load(p.c_str(), &W, &H, &hdr) != RT_OK);
        remove(p.c_str());
    }
    // camera + chapter 1
    float eye[3], cam[16];
    CHECK(rt_camera(30, 20, 4, eye, cam) == RT_OK);
    std::vector<double> img(32 * 24 * 3);
    CHECK(rt_chap1_render(32, 24, 2, 7u, img.data()) == RT_OK);
    std::vector<unsigned char> c8(32 * 24 * 3);
    CHECK(rt_chap1_to_rgb8(img.data(), 32, 24, c8.data()) == RT_OK);
    puts("host_sanitize: ok");
    return 0;
}
