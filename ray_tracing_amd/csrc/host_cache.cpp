// On-disk scene cache.  The reference rebuilds its BVH at every start (c5:843); at millions of triangles the
// bit-exact SAH build takes minutes.  The cache file holds exactly the arrays the reference uploads — Triangle_encoded[]
// (c5:54-64) and BVHNode_encoded[] (c5:65-70), raw little-endian fp32 — behind a small header with a caller-supplied key
// (a hash of whatever the arrays were built from), so a stale file is never used.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <vector>

#include "rt_host.h"

namespace {

struct CacheHeader {
    char magic[8];        // "RTAMDSC2"
    uint32_t tri_floats;  // 36
    uint32_t node_floats; // 12
    uint64_t key;
    uint64_t n_triangles;
    uint64_t n_nodes;
    uint64_t payload_hash;  // payload_hash() of the two arrays as stored
};
static_assert(sizeof(CacheHeader) == 48, "header layout is part of the file format");
const char kMagic[8] = {'R', 'T', 'A', 'M', 'D', 'S', 'C', '2'};

uint64_t fnv1a64(const void* data, size_t n, uint64_t h) {
    const unsigned char* p = (const unsigned char*)data;
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}

// The payload hash: FNV-1a 64 over the FNV-1a 64 hashes of the payload's 4 MiB blocks (triangles, then nodes; the last block
// of each array may be short).  A byte-serial hash over a 1.6 GB file is 1.6 s of one core at every start; the blocks are
// hashed by all cores (format version 2 — version-1 files are rejected as "not a scene cache file" and rebuilt).
constexpr size_t kHashBlock = 4u << 20;
uint64_t block_hashes(const void* data, size_t n, uint64_t h) {
    const size_t nb = (n + kHashBlock - 1) / kHashBlock;
    std::vector<uint64_t> bh(nb);
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t b = 0; b < nb; ++b) {
        const size_t off = b * kHashBlock, len = (n - off < kHashBlock) ? n - off : kHashBlock;
        bh[b] = fnv1a64((const unsigned char*)data + off, len, 0xcbf29ce484222325ull);
    }
    return fnv1a64(bh.data(), nb * sizeof(uint64_t), h);
}
uint64_t payload_hash(const void* tris, size_t tb, const void* nodes, size_t nb) {
    return block_hashes(nodes, nb, block_hashes(tris, tb, 0xcbf29ce484222325ull));
}

}  // namespace

using namespace rta;

extern "C" {

uint64_t rt_hash_fnv1a64(const void* data, uint64_t n_bytes, uint64_t seed) {
    return fnv1a64(data, (size_t)n_bytes, seed ? seed : 0xcbf29ce484222325ull);
}

int rt_scene_cache_save(const char* path, uint64_t key, const float* tris, int n_triangles, const float* nodes, int n_nodes) {
    if (!path || n_triangles < 0 || n_nodes < 0 || (n_triangles > 0 && !tris) || (n_nodes > 0 && !nodes))
        return set_error(RT_ERR_INVALID_ARG, "rt_scene_cache_save: bad arguments");
    CacheHeader h;
    memcpy(h.magic, kMagic, 8);
    h.tri_floats = 36, h.node_floats = 12, h.key = key;
    h.n_triangles = (uint64_t)n_triangles, h.n_nodes = (uint64_t)n_nodes;
    const size_t tb = (size_t)n_triangles * 36 * sizeof(float), nb = (size_t)n_nodes * 12 * sizeof(float);
    h.payload_hash = payload_hash(tris, tb, nodes, nb);
    // write a uniquely named file beside the target and rename it, so a reader never sees a half-written file and ranks
    // that hit a cold cache together (one process per GPU) do not write through each other
    char tmp[4096];
    if (snprintf(tmp, sizeof(tmp), "%s.tmp.XXXXXX", path) >= (int)sizeof(tmp)) return set_error(RT_ERR_INVALID_ARG, "rt_scene_cache_save: path too long");
    const int fd = mkstemp(tmp);
    FILE* f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
    if (!f) {
        if (fd >= 0) close(fd), remove(tmp);
        return set_error(RT_ERR_IO, "rt_scene_cache_save: cannot create %s", tmp);
    }
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    ok = ok && (tb == 0 || fwrite(tris, 1, tb, f) == tb);
    ok = ok && (nb == 0 || fwrite(nodes, 1, nb, f) == nb);
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp, path) != 0) {
        remove(tmp);
        return set_error(RT_ERR_IO, "rt_scene_cache_save: write to %s failed", path);
    }
    return RT_OK;
}

int rt_scene_cache_load(const char* path, uint64_t key, float** tris_out, int* n_triangles, float** nodes_out, int* n_nodes) {
    if (!path || !tris_out || !n_triangles || !nodes_out || !n_nodes)
        return set_error(RT_ERR_INVALID_ARG, "rt_scene_cache_load: NULL argument");
    *tris_out = *nodes_out = nullptr;
    *n_triangles = *n_nodes = 0;
    FILE* f = fopen(path, "rb");
    if (!f) return set_error(RT_ERR_IO, "rt_scene_cache_load: cannot open %s", path);
    CacheHeader h;
    float *t = nullptr, *n = nullptr;
    int rc = RT_OK;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, kMagic, 8) != 0 || h.tri_floats != 36 || h.node_floats != 12 ||
        h.n_triangles >= (1ull << 24) || h.n_nodes >= (1ull << 25)) {
        rc = set_error(RT_ERR_FORMAT, "rt_scene_cache_load: %s is not a scene cache file", path);
    } else if (h.key != key) {
        rc = set_error(RT_ERR_FORMAT, "rt_scene_cache_load: %s was built from other inputs (key %016llx, wanted %016llx)", path,
                       (unsigned long long)h.key, (unsigned long long)key);
    } else {
        const size_t tb = (size_t)h.n_triangles * 36 * sizeof(float), nb = (size_t)h.n_nodes * 12 * sizeof(float);
        t = (float*)malloc(tb ? tb : 1), n = (float*)malloc(nb ? nb : 1);
        if (!t || !n) {
            rc = set_error(RT_ERR_OOM, "rt_scene_cache_load: out of memory");
        } else if ((tb && fread(t, 1, tb, f) != tb) || (nb && fread(n, 1, nb, f) != nb) || fgetc(f) != EOF) {
            rc = set_error(RT_ERR_FORMAT, "rt_scene_cache_load: %s is truncated or has trailing bytes", path);
        } else if (payload_hash(t, tb, n, nb) != h.payload_hash) {
            rc = set_error(RT_ERR_FORMAT, "rt_scene_cache_load: %s is corrupt (payload hash mismatch)", path);
        }
    }
    fclose(f);
    if (rc != RT_OK) {
        free(t), free(n);
        return rc;
    }
    *tris_out = t, *nodes_out = n;
    *n_triangles = (int)h.n_triangles, *n_nodes = (int)h.n_nodes;
    return RT_OK;
}

}  // extern "C"
