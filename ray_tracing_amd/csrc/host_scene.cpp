// Host scene layer: OBJ readers, model transform, camera, HDR loader, HDR importance cache, tone-map, PNG writer.
// Produces, bit for bit, the arrays the reference uploads to the GPU (see include/ray_tracing_amd.h for citations).
// Compile with -ffp-contract=off.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <sstream>
#include <string>

#include "rt_host.h"

namespace rta {

// dst[i] += src[i] on all host cores (rt_render_multi's reduce: one call per replica, in replica order, so every element sees
// the additions in that order)
void add_into(float* dst, const float* src, size_t n) {
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) dst[i] += src[i];
}


static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- 4x4 matrices with GLM's operation order (type_mat4x4.inl) -------------------------------------------------

mat4 mat4_identity() {
    mat4 m;
    memset(&m, 0, sizeof(m));
    for (int i = 0; i < 4; ++i) m.c[i][i] = 1.0f;
    return m;
}

// Result[j] = ((A[0]*B[j][0] + A[1]*B[j][1]) + A[2]*B[j][2]) + A[3]*B[j][3]
mat4 mat4_mul(const mat4& a, const mat4& b) {
    mat4 r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r.c[j][i] = ((a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1]) + a.c[2][i] * b.c[j][2]) + a.c[3][i] * b.c[j][3];
    return r;
}

// (m[0]*v0 + m[1]*v1) + (m[2]*v2 + m[3]*v3)
void mat4_mul_vec4(const mat4& m, const float v[4], float out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = (m.c[0][i] * v[0] + m.c[1][i] * v[1]) + (m.c[2][i] * v[2] + m.c[3][i] * v[3]);
}

static float radians_(float deg) { return deg * 0.01745329251994329576923690768489f; }

// glm::rotate(m, angle, axis) (ext/matrix_transform.inl)
static mat4 mat4_rotate(const mat4& m, float angle, vec3 v) {
    const float c = cosf(angle), s = sinf(angle);
    const vec3 axis = normalize(v);
    const vec3 temp = (1.0f - c) * axis;
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = temp.x * axis.y + s * axis.z;
    R[0][2] = temp.x * axis.z - s * axis.y;
    R[1][0] = temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = temp.y * axis.z + s * axis.x;
    R[2][0] = temp.z * axis.x + s * axis.y;
    R[2][1] = temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    mat4 r;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 4; ++i) r.c[j][i] = (m.c[0][i] * R[j][0] + m.c[1][i] * R[j][1]) + m.c[2][i] * R[j][2];
    for (int i = 0; i < 4; ++i) r.c[3][i] = m.c[3][i];
    return r;
}

static mat4 mat4_scale(const mat4& m, vec3 v) {
    mat4 r;
    for (int i = 0; i < 4; ++i) {
        r.c[0][i] = m.c[0][i] * v.x;
        r.c[1][i] = m.c[1][i] * v.y;
        r.c[2][i] = m.c[2][i] * v.z;
        r.c[3][i] = m.c[3][i];
    }
    return r;
}

static mat4 mat4_translate(const mat4& m, vec3 v) {
    mat4 r = m;
    for (int i = 0; i < 4; ++i) r.c[3][i] = ((m.c[0][i] * v.x + m.c[1][i] * v.y) + m.c[2][i] * v.z) + m.c[3][i];
    return r;
}

// glm::lookAtRH
static mat4 mat4_lookat(vec3 eye, vec3 center, vec3 up) {
    const vec3 f = normalize(center - eye);
    const vec3 s = normalize(cross(f, up));
    const vec3 u = cross(s, f);
    mat4 r = mat4_identity();
    r.c[0][0] = s.x; r.c[1][0] = s.y; r.c[2][0] = s.z;
    r.c[0][1] = u.x; r.c[1][1] = u.y; r.c[2][1] = u.z;
    r.c[0][2] = -f.x; r.c[1][2] = -f.y; r.c[2][2] = -f.z;
    r.c[3][0] = -dot(s, eye);
    r.c[3][1] = -dot(u, eye);
    r.c[3][2] = dot(f, eye);
    return r;
}

// glm::inverse(mat4) (detail/func_matrix.inl, compute_inverse<4,4>)
static mat4 mat4_inverse(const mat4& M) {
    const float (*m)[4] = M.c;
    const float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    const float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    const float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    const float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    const float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    const float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    const float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    const float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    const float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    const float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    const float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    const float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    const float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    const float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    const float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    const float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    const float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    const float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    const float Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
    const float Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
    const float Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
    const float Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
    const float Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
    const float Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
    const float Vec0[4] = {m[1][0], m[0][0], m[0][0], m[0][0]};
    const float Vec1[4] = {m[1][1], m[0][1], m[0][1], m[0][1]};
    const float Vec2[4] = {m[1][2], m[0][2], m[0][2], m[0][2]};
    const float Vec3[4] = {m[1][3], m[0][3], m[0][3], m[0][3]};
    const float SignA[4] = {+1, -1, +1, -1};
    const float SignB[4] = {-1, +1, -1, +1};
    mat4 Inv;
    for (int i = 0; i < 4; ++i) {
        const float Inv0 = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i];
        const float Inv1 = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i];
        const float Inv2 = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i];
        const float Inv3 = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i];
        Inv.c[0][i] = Inv0 * SignA[i];
        Inv.c[1][i] = Inv1 * SignB[i];
        Inv.c[2][i] = Inv2 * SignA[i];
        Inv.c[3][i] = Inv3 * SignB[i];
    }
    const float Dot0[4] = {m[0][0] * Inv.c[0][0], m[0][1] * Inv.c[1][0], m[0][2] * Inv.c[2][0], m[0][3] * Inv.c[3][0]};
    const float Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
    const float OneOverDeterminant = 1.0f / Dot1;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) Inv.c[j][i] = Inv.c[j][i] * OneOverDeterminant;
    return Inv;
}

// ---- svpng-format PNG writer ------------------------------------------------------------------------------------

struct PngOut {
    std::vector<unsigned char> buf;
    uint32_t crc = 0;
    void put(unsigned u) { buf.push_back((unsigned char)(u & 255u)); }
    void u32(uint32_t u) { put(u >> 24); put(u >> 16); put(u >> 8); put(u); }
    void putc_crc(unsigned u) {
        static uint32_t table[256];
        static bool init = false;
        if (!init) {
            for (uint32_t i = 0; i < 256; ++i) {
                uint32_t c = i;
                for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
                table[i] = c;
            }
            init = true;
        }
        put(u);
        crc = table[(crc ^ u) & 255u] ^ (crc >> 8);
    }
    void u32c(uint32_t u) { putc_crc(u >> 24); putc_crc((u >> 16) & 255u); putc_crc((u >> 8) & 255u); putc_crc(u & 255u); }
    void begin(const char* tag, uint32_t len) {
        u32(len);
        crc = ~0u;
        for (int i = 0; i < 4; ++i) putc_crc((unsigned char)tag[i]);
    }
    void end() { u32(~crc); }
};

}  // namespace rta

using namespace rta;

extern "C" {

const char* rt_last_error(void) { return g_err; }

void rt_free(void* p) { free(p); }

void rt_material_default(rt_material* m) {
    if (!m) return;
    memset(m, 0, sizeof(*m));
    m->baseColor[0] = m->baseColor[1] = m->baseColor[2] = 1.0f;
    m->IOR = 1.0f;
}

int rt_mesh_create(rt_mesh** out) {
    if (!out) return set_error(RT_ERR_INVALID_ARG, "rt_mesh_create: out is NULL");
    *out = new rt_mesh();
    return RT_OK;
}

void rt_mesh_destroy(rt_mesh* m) { delete m; }

int rt_mesh_count(const rt_mesh* m) { return m ? (int)m->tris.size() : 0; }

const float* rt_mesh_data(const rt_mesh* m) { return (m && !m->tris.empty()) ? (const float*)m->tris.data() : nullptr; }

int rt_mesh_append(rt_mesh* m, const float* tris36, int n) {
    if (!m || n < 0 || (n > 0 && !tris36)) return set_error(RT_ERR_INVALID_ARG, "rt_mesh_append: bad arguments");
    const size_t off = m->tris.size();
    m->tris.resize(off + (size_t)n);
    if (n) memcpy((void*)&m->tris[off], tris36, (size_t)n * sizeof(TriRec));
    return RT_OK;
}

int rt_transform_matrix(const float rot[3], const float trans[3], const float scl[3], float out16[16]) {
    if (!rot || !trans || !scl || !out16) return set_error(RT_ERR_INVALID_ARG, "rt_transform_matrix: NULL argument");
    const mat4 unit = mat4_identity();
    const mat4 scale = mat4_scale(unit, v3(scl[0], scl[1], scl[2]));
    const mat4 translate = mat4_translate(unit, v3(trans[0], trans[1], trans[2]));
    mat4 rotate = unit;
    rotate = mat4_rotate(rotate, radians_(rot[0]), v3(1, 0, 0));
    rotate = mat4_rotate(rotate, radians_(rot[1]), v3(0, 1, 0));
    rotate = mat4_rotate(rotate, radians_(rot[2]), v3(0, 0, 1));
    const mat4 model = mat4_mul(mat4_mul(translate, rotate), scale);
    memcpy(out16, model.c, sizeof(float) * 16);
    return RT_OK;
}

int rt_mesh_read_obj(rt_mesh* mesh, const char* path, const rt_material* material, const float trans16[16], int smooth) {
    if (!mesh || !path || !material || !trans16) return set_error(RT_ERR_INVALID_ARG, "rt_mesh_read_obj: NULL argument");
    std::ifstream fin(path);
    if (!fin.is_open()) return set_error(RT_ERR_IO, "rt_mesh_read_obj: cannot open %s", path);
    mat4 trans;
    memcpy(trans.c, trans16, sizeof(float) * 16);

    std::vector<vec3> vertices;
    std::vector<unsigned> indices;
    float maxx = -11451419.19, maxy = -11451419.19, maxz = -11451419.19;
    float minx = 11451419.19, miny = 11451419.19, minz = 11451419.19;

    std::string line;
    long lineno = 0;
    while (std::getline(fin, line)) {
        ++lineno;
        std::istringstream sin(line);
        std::string type;
        float x = 0, y = 0, z = 0;
        int v0 = 0, v1 = 0, v2 = 0, vn0, vn1, vn2, vt0, vt1, vt2;
        char slash;
        int slashcnt = 0;
        for (size_t i = 0; i < line.size(); ++i)
            if (line[i] == '/') ++slashcnt;
        sin >> type;
        if (type == "v") {
            sin >> x >> y >> z;
            vertices.push_back(v3(x, y, z));
            maxx = fmax_(maxx, x), maxy = fmax_(maxy, y), maxz = fmax_(maxz, z);
            minx = fmin_(minx, x), miny = fmin_(miny, y), minz = fmin_(minz, z);
        }
        if (type == "f") {
            if (slashcnt == 6) {
                sin >> v0 >> slash >> vt0 >> slash >> vn0;
                sin >> v1 >> slash >> vt1 >> slash >> vn1;
                sin >> v2 >> slash >> vt2 >> slash >> vn2;
            } else if (slashcnt == 3) {
                sin >> v0 >> slash >> vt0;
                sin >> v1 >> slash >> vt1;
                sin >> v2 >> slash >> vt2;
            } else {
                sin >> v0 >> v1 >> v2;
            }
            if (sin.fail())
                return set_error(RT_ERR_FORMAT, "rt_mesh_read_obj: %s:%ld: unsupported face syntax", path, lineno);
            indices.push_back((unsigned)(v0 - 1));
            indices.push_back((unsigned)(v1 - 1));
            indices.push_back((unsigned)(v2 - 1));
        }
    }
    for (size_t i = 0; i < indices.size(); ++i)
        if (indices[i] >= vertices.size())
            return set_error(RT_ERR_FORMAT, "rt_mesh_read_obj: %s: face index %u out of range", path, indices[i] + 1);

    const float lenx = maxx - minx, leny = maxy - miny, lenz = maxz - minz;
    const float maxaxis = fmax_(lenx, fmax_(leny, lenz));
    for (auto& v : vertices) {
        v.x /= maxaxis;
        v.y /= maxaxis;
        v.z /= maxaxis;
    }
    for (auto& v : vertices) {
        const float in[4] = {v.x, v.y, v.z, 1.0f};
        float o[4];
        mat4_mul_vec4(trans, in, o);
        v = v3(o[0], o[1], o[2]);
    }

    std::vector<vec3> normals(vertices.size(), v3(0, 0, 0));
    for (size_t i = 0; i < indices.size(); i += 3) {
        const vec3 p1 = vertices[indices[i]], p2 = vertices[indices[i + 1]], p3 = vertices[indices[i + 2]];
        const vec3 n = normalize(cross(p2 - p1, p3 - p1));
        normals[indices[i]] += n;
        normals[indices[i + 1]] += n;
        normals[indices[i + 2]] += n;
    }

    const size_t offset = mesh->tris.size();
    mesh->tris.resize(offset + indices.size() / 3);
    for (size_t i = 0; i < indices.size(); i += 3) {
        TriRec& t = mesh->tris[offset + i / 3];
        t.p1 = vertices[indices[i]];
        t.p2 = vertices[indices[i + 1]];
        t.p3 = vertices[indices[i + 2]];
        if (!smooth) {
            const vec3 n = normalize(cross(t.p2 - t.p1, t.p3 - t.p1));
            t.n1 = n, t.n2 = n, t.n3 = n;
        } else {
            t.n1 = normalize(normals[indices[i]]);
            t.n2 = normalize(normals[indices[i + 1]]);
            t.n3 = normalize(normals[indices[i + 2]]);
        }
        t.m = *material;
    }
    return RT_OK;
}

int rt_mesh_read_obj_chap2(rt_mesh* mesh, const char* path, float scale, float y_offset) {
    if (!mesh || !path) return set_error(RT_ERR_INVALID_ARG, "rt_mesh_read_obj_chap2: NULL argument");
    std::ifstream fin(path);
    if (!fin.is_open()) return set_error(RT_ERR_IO, "rt_mesh_read_obj_chap2: cannot open %s", path);
    std::vector<vec3> vertices;
    std::vector<unsigned> indices;
    std::string line;
    while (std::getline(fin, line)) {
        std::istringstream sin(line);
        std::string type;
        float x = 0, y = 0, z = 0;
        int v0 = 0, v1 = 0, v2 = 0;
        sin >> type;
        if (type == "v") {
            sin >> x >> y >> z;
            vertices.push_back(v3(x, y, z));
        }
        if (type == "f") {
            sin >> v0 >> v1 >> v2;
            indices.push_back((unsigned)(v0 - 1));
            indices.push_back((unsigned)(v1 - 1));
            indices.push_back((unsigned)(v2 - 1));
        }
    }
    for (size_t i = 0; i < indices.size(); ++i)
        if (indices[i] >= vertices.size())
            return set_error(RT_ERR_FORMAT, "rt_mesh_read_obj_chap2: %s: face index out of range", path);
    for (auto& v : vertices) {
        v.x *= scale, v.y *= scale, v.z *= scale;
        v.y -= y_offset;
    }
    rt_material dm;
    rt_material_default(&dm);
    const size_t offset = mesh->tris.size();
    mesh->tris.resize(offset + indices.size() / 3);
    for (size_t i = 0; i < indices.size(); i += 3) {
        TriRec& t = mesh->tris[offset + i / 3];
        t.p1 = vertices[indices[i]];
        t.p2 = vertices[indices[i + 1]];
        t.p3 = vertices[indices[i + 2]];
        const vec3 n = normalize(cross(t.p2 - t.p1, t.p3 - t.p1));
        t.n1 = n, t.n2 = n, t.n3 = n;
        t.m = dm;
    }
    return RT_OK;
}

int rt_camera(float rotat_deg, float up_deg, float r, float eye_out[3], float cam16[16]) {
    if (!eye_out || !cam16) return set_error(RT_ERR_INVALID_ARG, "rt_camera: NULL argument");
    vec3 eye = v3(-sinf(radians_(rotat_deg)) * cosf(radians_(up_deg)), sinf(radians_(up_deg)),
                  cosf(radians_(rotat_deg)) * cosf(radians_(up_deg)));
    eye.x *= r, eye.y *= r, eye.z *= r;
    const mat4 cam = mat4_inverse(mat4_lookat(eye, v3(0, 0, 0), v3(0, 1, 0)));
    eye_out[0] = eye.x, eye_out[1] = eye.y, eye_out[2] = eye.z;
    memcpy(cam16, cam.c, sizeof(float) * 16);
    return RT_OK;
}

// ---- Radiance .hdr ----------------------------------------------------------------------------------------------

typedef unsigned char RGBE4[4];

static bool hdr_old_decrunch(RGBE4* scanline, int len, FILE* file, RGBE4* first) {
    int rshift = 0;
    while (len > 0) {
        scanline[0][0] = (unsigned char)fgetc(file);
        scanline[0][1] = (unsigned char)fgetc(file);
        scanline[0][2] = (unsigned char)fgetc(file);
        scanline[0][3] = (unsigned char)fgetc(file);
        if (feof(file)) return false;
        if (scanline[0][0] == 1 && scanline[0][1] == 1 && scanline[0][2] == 1) {
            if (scanline == first) return false;  // repeat of a pixel before the scanline: malformed
            if (rshift > 24) return false;        // a fifth consecutive repeat marker: the count would shift past an int
            // (the count is formed in 64 bits: 255 << 24 does not fit a signed int)
            for (long long i = (long long)scanline[0][3] << rshift; i > 0 && len > 0; i--) {
                memcpy(&scanline[0][0], &scanline[-1][0], 4);
                scanline++;
                len--;
            }
            rshift += 8;
        } else {
            scanline++;
            len--;
            rshift = 0;
        }
    }
    return true;
}

static bool hdr_decrunch(RGBE4* scanline, int len, FILE* file) {
    if (len < 8 || len > 0x7fff) return hdr_old_decrunch(scanline, len, file, scanline);
    int i = fgetc(file);
    if (i != 2) {
        fseek(file, -1, SEEK_CUR);
        return hdr_old_decrunch(scanline, len, file, scanline);
    }
    scanline[0][1] = (unsigned char)fgetc(file);
    scanline[0][2] = (unsigned char)fgetc(file);
    i = fgetc(file);
    if (scanline[0][1] != 2 || (scanline[0][2] & 128)) {
        scanline[0][0] = 2;
        scanline[0][3] = (unsigned char)i;
        return hdr_old_decrunch(scanline + 1, len - 1, file, scanline);
    }
    for (i = 0; i < 4; i++) {
        for (int j = 0; j < len;) {
            unsigned char code = (unsigned char)fgetc(file);
            if (feof(file)) return false;
            if (code > 128) {
                code &= 127;
                const unsigned char val = (unsigned char)fgetc(file);
                while (code-- && j < len) scanline[j++][i] = val;
            } else {
                while (code-- && j < len) scanline[j++][i] = (unsigned char)fgetc(file);
            }
        }
    }
    return feof(file) ? false : true;
}

int rt_hdr_load(const char* path, int* width, int* height, float** rgb) {
    if (!path || !width || !height || !rgb) return set_error(RT_ERR_INVALID_ARG, "rt_hdr_load: NULL argument");
    FILE* file = fopen(path, "rb");
    if (!file) return set_error(RT_ERR_IO, "rt_hdr_load: cannot open %s", path);
    char str[16];
    if (fread(str, 10, 1, file) != 1 || memcmp(str, "#?RADIANCE", 10)) {
        fclose(file);
        return set_error(RT_ERR_FORMAT, "rt_hdr_load: %s is not a Radiance file", path);
    }
    fseek(file, 1, SEEK_CUR);
    int c = 0, oldc;
    while (true) {  // header commands up to the empty line
        oldc = c;
        c = fgetc(file);
        if (c == EOF) {
            fclose(file);
            return set_error(RT_ERR_FORMAT, "rt_hdr_load: %s: truncated header", path);
        }
        if (c == 0xa && oldc == 0xa) break;
    }
    char reso[200];
    int i = 0;
    while (true) {
        c = fgetc(file);
        if (c == EOF || i >= 198) {
            fclose(file);
            return set_error(RT_ERR_FORMAT, "rt_hdr_load: %s: bad resolution line", path);
        }
        reso[i++] = (char)c;
        if (c == 0xa) break;
    }
    reso[i] = 0;
    int w = 0, h = 0;
    if (sscanf(reso, "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || (long long)w * h > (1ll << 28)) {
        fclose(file);
        return set_error(RT_ERR_FORMAT, "rt_hdr_load: %s: unsupported resolution line", path);
    }
    float* cols = (float*)calloc((size_t)w * h * 3, sizeof(float));
    RGBE4* scanline = (RGBE4*)calloc((size_t)w, sizeof(RGBE4));
    if (!cols || !scanline) {
        free(cols), free(scanline), fclose(file);
        return set_error(RT_ERR_OOM, "rt_hdr_load: out of memory");
    }
    float* out = cols;
    for (int y = h - 1; y >= 0; y--) {
        if (!hdr_decrunch(scanline, w, file)) break;  // hdr:87-88: a short file leaves the remaining rows black
        for (int x = 0; x < w; ++x) {
            const int expo = scanline[x][3] - 128;
            const float d = (float)pow(2, expo);  // hdr:99-104
            out[0] = (scanline[x][0] / 256.0f) * d;
            out[1] = (scanline[x][1] / 256.0f) * d;
            out[2] = (scanline[x][2] / 256.0f) * d;
            out += 3;
        }
    }
    free(scanline);
    fclose(file);
    *width = w, *height = h, *rgb = cols;
    return RT_OK;
}

// calculateHdrCache (c5:595-698) on flat arrays.  Every floating-point operation and its order is the reference's:
// double-precision luminance weights rounded to float, one fp32 running sum in row-major order, column-major prefix sums.
int rt_hdr_cache(const float* HDR, int width, int height, float* cache) {
    if (!HDR || !cache || width <= 0 || height <= 0) return set_error(RT_ERR_INVALID_ARG, "rt_hdr_cache: bad arguments");
    const size_t W = (size_t)width, H = (size_t)height;
    std::vector<float> pdf(W * H);
    float lumSum = 0.0;
    for (size_t i = 0; i < H; i++)
        for (size_t j = 0; j < W; j++) {
            const float R = HDR[3 * (i * W + j)], G = HDR[3 * (i * W + j) + 1], B = HDR[3 * (i * W + j) + 2];
            const float lum = 0.2 * R + 0.7 * G + 0.1 * B;  // evaluated in double, as written at c5:607
            pdf[i * W + j] = lum;
            lumSum += lum;
        }
    for (size_t k = 0; k < W * H; ++k) pdf[k] /= lumSum;

    std::vector<float> pdf_x_margin(W, 0.0f);
    for (size_t j = 0; j < W; j++)
        for (size_t i = 0; i < H; i++) pdf_x_margin[j] += pdf[i * W + j];
    std::vector<float> cdf_x_margin = pdf_x_margin;
    for (size_t i = 1; i < W; i++) cdf_x_margin[i] += cdf_x_margin[i - 1];

    // cdf_y[j*H + i]: conditional CDF of row i given column j (the reference's transposed cdf_y_condiciton)
    std::vector<float> cdf_y(W * H);
    for (size_t j = 0; j < W; j++) {
        float acc = 0;
        for (size_t i = 0; i < H; i++) {
            const float p = pdf[i * W + j] / pdf_x_margin[j];
            acc = (i == 0) ? p : (p + acc);  // c5:643: cdf[i] += cdf[i-1]  ==  p_i + cdf[i-1]
            cdf_y[j * H + i] = acc;
        }
    }
    for (size_t j = 0; j < W; ++j) {
        const float xi_2 = float(j) / width;
        for (size_t i = 0; i < H; ++i) {
            const float xi_1 = float(i) / height;
            const int x = (int)(std::lower_bound(cdf_x_margin.begin(), cdf_x_margin.end(), xi_1) - cdf_x_margin.begin());
            // c5:676 indexes cdf_y_condiciton[x] with x == width when xi_1 exceeds the last CDF entry (UB in the
            // reference); clamp the column so the table stays defined.
            const size_t xc = (x < width) ? (size_t)x : W - 1;
            const float* col = &cdf_y[xc * H];
            const int y = (int)(std::lower_bound(col, col + H, xi_2) - col);
            cache[3 * (i * W + j)] = float(x) / width;
            cache[3 * (i * W + j) + 1] = float(y) / height;
            cache[3 * (i * W + j) + 2] = pdf[i * W + j];
        }
    }
    return RT_OK;
}

// pass3.fsh:14-25 followed by the fixed-function float -> unorm8 conversion (clamp, *255, round to nearest).
int rt_tonemap_rgb8(const float* rgb, int width, int height, unsigned char* out, int flip_y) {
    if (!rgb || !out || width <= 0 || height <= 0) return set_error(RT_ERR_INVALID_ARG, "rt_tonemap_rgb8: bad arguments");
    for (int y = 0; y < height; ++y) {
        const float* src = rgb + (size_t)y * width * 3;
        unsigned char* dst = out + (size_t)(flip_y ? (height - 1 - y) : y) * width * 3;
        for (int x = 0; x < width; ++x) {
            const float r = src[3 * x], g = src[3 * x + 1], b = src[3 * x + 2];
            const float luminance = (0.3f * r + 0.6f * g) + 0.1f * b;
            const float k = 1.0f + luminance / 1.5f;
            const float c[3] = {r * 1.0f / k, g * 1.0f / k, b * 1.0f / k};
            for (int ch = 0; ch < 3; ++ch) {
                float v = powf(c[ch], 1.0f / 2.2f);
                v = (v >= 0.0f) ? ((v <= 1.0f) ? v : 1.0f) : 0.0f;  // NaN -> 0
                dst[3 * x + ch] = (unsigned char)(int)floorf(v * 255.0f + 0.5f);
            }
        }
    }
    return RT_OK;
}

int rt_write_png(const char* path, int width, int height, const unsigned char* img, int alpha) {
    if (!path || !img || width <= 0 || height <= 0) return set_error(RT_ERR_INVALID_ARG, "rt_write_png: bad arguments");
    const unsigned w = (unsigned)width, h = (unsigned)height;
    const unsigned p = w * (alpha ? 4u : 3u) + 1u;  // pitch incl. filter byte
    if (p > 0xffffu) return set_error(RT_ERR_INVALID_ARG, "rt_write_png: row of %u bytes exceeds a stored deflate block", p);
    PngOut o;
    o.buf.reserve((size_t)h * (p + 5) + 64);
    const unsigned char magic[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    for (int i = 0; i < 8; ++i) o.put(magic[i]);
    o.begin("IHDR", 13);
    o.u32c(w), o.u32c(h);
    o.putc_crc(8), o.putc_crc(alpha ? 6 : 2);
    o.putc_crc(0), o.putc_crc(0), o.putc_crc(0);
    o.end();
    o.begin("IDAT", 2 + h * (5 + p) + 4);
    o.putc_crc(0x78), o.putc_crc(1);
    uint32_t a = 1, b = 0;
    for (unsigned y = 0; y < h; y++) {
        o.putc_crc(y == h - 1);
        o.putc_crc(p & 255u), o.putc_crc((p >> 8) & 255u);
        o.putc_crc((~p) & 255u), o.putc_crc(((~p) >> 8) & 255u);
        o.putc_crc(0);
        a = (a + 0) % 65521u, b = (b + a) % 65521u;
        for (unsigned x = 0; x < p - 1; x++, img++) {
            o.putc_crc(*img);
            a = (a + *img) % 65521u, b = (b + a) % 65521u;
        }
    }
    o.u32c((b << 16) | a);
    o.end();
    o.begin("IEND", 0);
    o.end();
    FILE* fp = fopen(path, "wb");
    if (!fp) return set_error(RT_ERR_IO, "rt_write_png: cannot open %s", path);
    const size_t n = fwrite(o.buf.data(), 1, o.buf.size(), fp);
    fclose(fp);
    if (n != o.buf.size()) return set_error(RT_ERR_IO, "rt_write_png: short write to %s", path);
    return RT_OK;
}

}  // extern "C"
