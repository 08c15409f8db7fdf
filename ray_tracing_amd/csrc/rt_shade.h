// Shading side of the path loop: RNG, Sobol, Disney BRDF evaluate/sample/pdf, environment lookups, MIS.
// Host+device inline code; the HIP kernels in rt_device.hip are the only product callers.
// Operation order follows path_tracing_{3,4,5}chap/shaders/fshader.fsh (f3/f4/f5) so results are reproducible
// to the bit by the CPU oracle.  Compile with -ffp-contract=off.
#pragma once
#include "rt_detmath.h"
#include "rt_vec.h"

namespace rta {

constexpr float PI_ = 3.1415926f;  // #define PI 3.1415926, f5:27
constexpr float INF_ = 114514.0f;  // #define INF 114514.0, f5:28

// ---- R1: seed / wang_hash / rand, f5:320-336 -------------------------------------------------------------------
RT_HD unsigned wang_hash(unsigned& seed) {
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}
RT_HD float rand01(unsigned& seed) { return (float)wang_hash(seed) / 4294967296.0f; }

RT_HD unsigned pixel_seed(float pixx, float pixy, int width, int height, unsigned frame) {
    const unsigned px = (unsigned)((pixx * 0.5f + 0.5f) * (float)width);
    const unsigned py = (unsigned)((pixy * 0.5f + 0.5f) * (float)height);
    return (px * 1973u + py * 9277u + frame * 26699u) | 1u;
}

// ---- R2: Sobol, f5:338-365.  V[] is the direction-number table of f5:339 (data). -------------------------------
RT_HD float sobol_dim(const unsigned* __restrict__ V, int d, unsigned gray) {
    unsigned result = 0u;
    const int offset = 32 * d;
    for (int j = 0; gray != 0u; gray >>= 1, j++)
        if (gray & 1u) result ^= V[offset + j];
    return (float)result * (1.0f / (float)0xFFFFFFFFu);
}

// ---- R3: Cranley-Patterson rotation, f5:367-383 ------------------------------------------------------------------
RT_HD void cp_shift(float pixx, float pixy, int width, int height, float* u, float* v) {
    unsigned pseed = pixel_seed(pixx, pixy, width, height, (unsigned)(114514 / 1919));
    *u = (float)wang_hash(pseed) / 4294967296.0f;
    *v = (float)wang_hash(pseed) / 4294967296.0f;
}
RT_HD float cp_rotate(float p, float shift) {
    p += shift;
    if (p > 1.0f) p -= 1.0f;
    if (p < 0.0f) p += 1.0f;
    return p;
}

// ---- material: texels 6..11 of Triangle_encoded, f5:118-143 ------------------------------------------------------
struct Mat {
    vec3 emissive, baseColor;
    float subsurface, metallic, specular;
    float specularTint, roughness, anisotropic;
    float sheen, sheenTint, clearcoat;
    float clearcoatGloss;
};

// ---- F1: helper terms, f5:387-428 --------------------------------------------------------------------------------
RT_HD float sqr(float x) { return x * x; }
RT_HD float SchlickFresnel(float u) {
    const float m = clamp_(1.0f - u, 0.0f, 1.0f);
    const float m2 = m * m;
    return m2 * m2 * m;
}
RT_HD float GTR1(float NdotH, float a) {
    if (a >= 1.0f) return 1.0f / PI_;
    const float a2 = a * a;
    const float t = 1.0f + (a2 - 1.0f) * NdotH * NdotH;
    return (a2 - 1.0f) / (PI_ * det_log(a2) * t);
}
RT_HD float GTR2(float NdotH, float a) {
    const float a2 = a * a;
    const float t = 1.0f + (a2 - 1.0f) * NdotH * NdotH;
    return a2 / (PI_ * t * t);
}
RT_HD float GTR2_aniso(float NdotH, float HdotX, float HdotY, float ax, float ay) {
    return 1.0f / (PI_ * ax * ay * sqr(sqr(HdotX / ax) + sqr(HdotY / ay) + NdotH * NdotH));
}
RT_HD float smithG_GGX(float NdotV, float alphaG) {
    const float a = alphaG * alphaG;
    const float b = NdotV * NdotV;
    return 1.0f / (NdotV + sqrtf(a + b - a * b));
}
RT_HD float smithG_GGX_aniso(float NdotV, float VdotX, float VdotY, float ax, float ay) {
    return 1.0f / (NdotV + sqrtf(sqr(VdotX * ax) + sqr(VdotY * ay) + sqr(NdotV)));
}

// ---- S2: tangent frames, f5:486-501 -------------------------------------------------------------------------------
RT_HD vec3 toNormalHemisphere(vec3 v, vec3 N) {
    vec3 helper = v3(1, 0, 0);
    if (fabs_(N.x) > 0.999f) helper = v3(0, 0, 1);
    const vec3 tangent = normalize(cross(N, helper));
    const vec3 bitangent = normalize(cross(N, tangent));
    return v.x * tangent + v.y * bitangent + v.z * N;
}
RT_HD void getTagent(vec3 N, vec3& tangent, vec3& bitangent) {  // f4:139-145 (names as in the reference)
    vec3 helper = v3(1, 0, 0);
    if (fabs_(N.x) > 0.999f) helper = v3(0, 0, 1);
    bitangent = normalize(cross(N, helper));
    tangent = normalize(cross(N, bitangent));
}

// ---- F2 / F2' / F3: Disney BRDF value and sampling pdf ------------------------------------------------------------
// ANISO = false: f5:430-482 (chap5);  ANISO = true: f4:214-278 (chap4, X = tangent, Y = bitangent of getTagent).
// WANT_PDF additionally returns BRDF_Pdf (f5:662-699), which shares H, NdotH, Ds, Dr with the evaluation.
template <bool ANISO, bool WANT_PDF>
RT_HD vec3 disney_eval(vec3 V, vec3 N, vec3 L, vec3 X, vec3 Y, const Mat& m, float* pdf_out) {
    const float NdotL = dot(N, L);
    const float NdotV = dot(N, V);
    if (NdotL < 0.0f || NdotV < 0.0f) {
        if (WANT_PDF) *pdf_out = 0.0f;
        return v3(0);
    }
    const vec3 H = normalize(L + V);
    const float NdotH = dot(N, H);
    const float LdotH = dot(L, H);

    const vec3 Cdlin = m.baseColor;
    const float Cdlum = 0.3f * Cdlin.x + 0.6f * Cdlin.y + 0.1f * Cdlin.z;
    const vec3 Ctint = (Cdlum > 0.0f) ? (Cdlin / Cdlum) : v3(1);
    const vec3 Cspec = m.specular * mix3(v3(1), Ctint, m.specularTint);
    const vec3 Cspec0 = mix3(0.08f * Cspec, Cdlin, m.metallic);
    const vec3 Csheen = mix3(v3(1), Ctint, m.sheenTint);

    const float Fd90 = 0.5f + 2.0f * m.roughness * LdotH * LdotH;
    const float FL = SchlickFresnel(NdotL);
    const float FV = SchlickFresnel(NdotV);
    const float Fd = mix_(1.0f, Fd90, FL) * mix_(1.0f, Fd90, FV);
    const float Fss90 = LdotH * LdotH * m.roughness;
    const float Fss = mix_(1.0f, Fss90, FL) * mix_(1.0f, Fss90, FV);
    const float ss = 1.25f * (Fss * (1.0f / (NdotL + NdotV) - 0.5f) + 0.5f);

    float Ds, Gs;
    const float FH = SchlickFresnel(LdotH);
    const vec3 Fs = mix3(Cspec0, v3(1), FH);
    if (ANISO) {
        const float aspect = sqrtf(1.0f - m.anisotropic * 0.9f);
        const float ax = fmax_(0.001f, sqr(m.roughness) / aspect);
        const float ay = fmax_(0.001f, sqr(m.roughness) * aspect);
        Ds = GTR2_aniso(NdotH, dot(H, X), dot(H, Y), ax, ay);
        Gs = smithG_GGX_aniso(NdotL, dot(L, X), dot(L, Y), ax, ay);
        Gs *= smithG_GGX_aniso(NdotV, dot(V, X), dot(V, Y), ax, ay);
    } else {
        const float alpha = fmax_(0.001f, sqr(m.roughness));
        Ds = GTR2(NdotH, alpha);
        Gs = smithG_GGX(NdotL, m.roughness);
        Gs *= smithG_GGX(NdotV, m.roughness);
    }
    const float Dr = GTR1(NdotH, mix_(0.1f, 0.001f, m.clearcoatGloss));
    const float Fr = mix_(0.04f, 1.0f, FH);
    const float Gr = smithG_GGX(NdotL, 0.25f) * smithG_GGX(NdotV, 0.25f);

    const vec3 Fsheen = FH * m.sheen * Csheen;
    const vec3 diffuse = (1.0f / PI_) * mix_(Fd, ss, m.subsurface) * Cdlin + Fsheen;
    const vec3 specular = Gs * Fs * Ds;
    const vec3 clearcoat = v3(0.25f * Gr * Fr * Dr * m.clearcoat);

    if (WANT_PDF) {  // f5:672-698 (always the isotropic Ds)
        const float pdf_diffuse = NdotL / PI_;
        const float pdf_specular = Ds * NdotH / (4.0f * LdotH);
        const float pdf_clearcoat = Dr * NdotH / (4.0f * LdotH);
        const float r_diffuse = (1.0f - m.metallic);
        const float r_specular = 1.0f;
        const float r_clearcoat = 0.25f * m.clearcoat;
        const float r_sum = r_diffuse + r_specular + r_clearcoat;
        const float p_diffuse = r_diffuse / r_sum;
        const float p_specular = r_specular / r_sum;
        const float p_clearcoat = r_clearcoat / r_sum;
        const float pdf = p_diffuse * pdf_diffuse + p_specular * pdf_specular + p_clearcoat * pdf_clearcoat;
        *pdf_out = fmax_(1e-10f, pdf);
    }
    return diffuse * (1.0f - m.metallic) + specular + clearcoat;
}

// ---- chapter 5 with the material-only terms taken out -----------------------------------------------------------------
// Many terms of BRDF_Evaluate / BRDF_Pdf / SampleBRDF (f5:430-482, 662-699, 575-608) depend on the material alone.  The
// device layer computes them once per triangle at scene creation (mat_precompute: the very expressions of disney_eval /
// SampleBRDF above, fp32, same order) and the path loop reads them back: bit-identical values, ~200 fewer instructions per
// shade pass (three vec3 / float quotients, a log, three weight quotients, several mixes).
struct MatPre {
    vec3 Cspec0, Csheen;
    float a2, a2m1;                      // GTR2: alpha^2 and alpha^2 - 1 with alpha = max(0.001, roughness^2)
    float g1_a, g1_a2, g1_a2m1, g1_k;    // GTR1: a = mix(0.1, 0.001, clearcoatGloss), a^2, a^2 - 1, PI * log(a^2)
    float g1_lg;                         // log(a^2): the inner log of pow(a^2, 1 - xi) in the clearcoat lobe's sampler
    float p_d, p_s, p_c;                 // lobe weights r / r_sum
};
RT_HD MatPre mat_precompute(const Mat& m) {
    MatPre q;
    const vec3 Cdlin = m.baseColor;
    const float Cdlum = 0.3f * Cdlin.x + 0.6f * Cdlin.y + 0.1f * Cdlin.z;
    const vec3 Ctint = (Cdlum > 0.0f) ? (Cdlin / Cdlum) : v3(1);
    const vec3 Cspec = m.specular * mix3(v3(1), Ctint, m.specularTint);
    q.Cspec0 = mix3(0.08f * Cspec, Cdlin, m.metallic);
    q.Csheen = mix3(v3(1), Ctint, m.sheenTint);
    const float alpha = fmax_(0.001f, sqr(m.roughness));
    q.a2 = alpha * alpha;
    q.a2m1 = q.a2 - 1.0f;
    q.g1_a = mix_(0.1f, 0.001f, m.clearcoatGloss);
    q.g1_a2 = q.g1_a * q.g1_a;
    q.g1_a2m1 = q.g1_a2 - 1.0f;
    q.g1_lg = det_log(q.g1_a2);
    q.g1_k = PI_ * q.g1_lg;
    const float r_diffuse = (1.0f - m.metallic);
    const float r_specular = 1.0f;
    const float r_clearcoat = 0.25f * m.clearcoat;
    const float r_sum = r_diffuse + r_specular + r_clearcoat;
    q.p_d = r_diffuse / r_sum;
    q.p_s = r_specular / r_sum;
    q.p_c = r_clearcoat / r_sum;
    return q;
}
// BRDF_Evaluate + BRDF_Pdf of chapter 5 (= disney_eval<false, true>) on the precomputed terms.  `m` needs baseColor,
// roughness, subsurface, sheen, metallic, clearcoat only.
RT_HD vec3 disney_eval5_pre(vec3 V, vec3 N, vec3 L, const Mat& m, const MatPre& q, float* pdf_out) {
    const float NdotL = dot(N, L);
    const float NdotV = dot(N, V);
    if (NdotL < 0.0f || NdotV < 0.0f) {
        *pdf_out = 0.0f;
        return v3(0);
    }
    const vec3 H = normalize(L + V);
    const float NdotH = dot(N, H);
    const float LdotH = dot(L, H);
    const vec3 Cdlin = m.baseColor;

    const float Fd90 = 0.5f + 2.0f * m.roughness * LdotH * LdotH;
    const float FL = SchlickFresnel(NdotL);
    const float FV = SchlickFresnel(NdotV);
    const float Fd = mix_(1.0f, Fd90, FL) * mix_(1.0f, Fd90, FV);
    const float Fss90 = LdotH * LdotH * m.roughness;
    const float Fss = mix_(1.0f, Fss90, FL) * mix_(1.0f, Fss90, FV);
    const float ss = 1.25f * (Fss * (1.0f / (NdotL + NdotV) - 0.5f) + 0.5f);

    const float FH = SchlickFresnel(LdotH);
    const vec3 Fs = mix3(q.Cspec0, v3(1), FH);
    const float t2 = 1.0f + q.a2m1 * NdotH * NdotH;  // GTR2
    const float Ds = q.a2 / (PI_ * t2 * t2);
    float Gs = smithG_GGX(NdotL, m.roughness);
    Gs *= smithG_GGX(NdotV, m.roughness);
    float Dr;  // GTR1
    if (q.g1_a >= 1.0f) {
        Dr = 1.0f / PI_;
    } else {
        const float t1 = 1.0f + q.g1_a2m1 * NdotH * NdotH;
        Dr = q.g1_a2m1 / (q.g1_k * t1);
    }
    const float Fr = mix_(0.04f, 1.0f, FH);
    const float Gr = smithG_GGX(NdotL, 0.25f) * smithG_GGX(NdotV, 0.25f);

    const vec3 Fsheen = FH * m.sheen * q.Csheen;
    const vec3 diffuse = (1.0f / PI_) * mix_(Fd, ss, m.subsurface) * Cdlin + Fsheen;
    const vec3 specular = Gs * Fs * Ds;
    const vec3 clearcoat = v3(0.25f * Gr * Fr * Dr * m.clearcoat);

    const float pdf_diffuse = NdotL / PI_;
    const float pdf_specular = Ds * NdotH / (4.0f * LdotH);
    const float pdf_clearcoat = Dr * NdotH / (4.0f * LdotH);
    const float pdf = q.p_d * pdf_diffuse + q.p_s * pdf_specular + q.p_c * pdf_clearcoat;
    *pdf_out = fmax_(1e-10f, pdf);
    return diffuse * (1.0f - m.metallic) + specular + clearcoat;
}

// ---- S1/S3/S4/S5: direction sampling, f5:509-608 -------------------------------------------------------------------
RT_HD vec3 SampleHemisphereRand(unsigned& seed) {  // f4:110-115: z then phi
    const float z = rand01(seed);
    const float r = fmax_(0.0f, sqrtf(1.0f - z * z));
    const float phi = 2.0f * PI_ * rand01(seed);
    return v3(r * det_cos(phi), r * det_sin(phi), z);
}
RT_HD vec3 SampleCosineHemisphere(float xi_1, float xi_2, vec3 N) {
    const float r = sqrtf(xi_1);
    const float theta = xi_2 * 2.0f * PI_;
    const float x = r * det_cos(theta);
    const float y = r * det_sin(theta);
    const float z = sqrtf(1.0f - x * x - y * y);
    return toNormalHemisphere(v3(x, y, z), N);
}
RT_HD vec3 SampleGTR(float xi_1, float cos_theta_h, vec3 V, vec3 N) {  // common tail of f5:537-573
    const float phi_h = 2.0f * PI_ * xi_1;
    const float sin_phi_h = det_sin(phi_h);
    const float cos_phi_h = det_cos(phi_h);
    const float sin_theta_h = sqrtf(fmax_(0.0f, 1.0f - cos_theta_h * cos_theta_h));
    vec3 H = v3(sin_theta_h * cos_phi_h, sin_theta_h * sin_phi_h, cos_theta_h);
    H = toNormalHemisphere(H, N);
    return reflect(-V, H);
}
// f5:575-608.  The three lobes (cosine hemisphere f5:524-535, GTR2 f5:556-573, GTR1 f5:537-554) are evaluated through ONE
// shared sin/cos, square root, quotient and toNormalHemisphere, with the lobe selecting the operands: per lane every value
// goes through exactly the operations of its own lobe's formula (SampleCosineHemisphere / SampleGTR above), so the result is
// bit-identical to the branchy form, but a wave whose lanes picked different lobes runs the common part once, not three times.
RT_HD vec3 SampleBRDF(float xi_1, float xi_2, float xi_3, vec3 V, vec3 N, const Mat& m) {
    const float alpha_GTR1 = mix_(0.1f, 0.001f, m.clearcoatGloss);
    const float alpha_GTR2 = fmax_(0.001f, sqr(m.roughness));
    const float r_diffuse = (1.0f - m.metallic);
    const float r_specular = 1.0f;
    const float r_clearcoat = 0.25f * m.clearcoat;
    const float r_sum = r_diffuse + r_specular + r_clearcoat;
    const float p_diffuse = r_diffuse / r_sum;
    const float p_specular = r_specular / r_sum;
    const float rd = xi_3;
    const bool diffuse = rd <= p_diffuse;
    const bool specular = !diffuse && (p_diffuse < rd && rd <= p_diffuse + p_specular);
    const bool clearcoat = !diffuse && !specular && (p_diffuse + p_specular < rd);
    if (!(diffuse || specular || clearcoat)) return v3(0, 1, 0);  // NaN weights
    // cos(theta_h) of the two GTR lobes: sqrt(num / den)
    float num = 1.0f - xi_2, den = 1.0f + (alpha_GTR2 * alpha_GTR2 - 1.0f) * xi_2;
    if (clearcoat) {
        num = 1.0f - det_pow(alpha_GTR1 * alpha_GTR1, 1.0f - xi_2);
        den = 1.0f - alpha_GTR1 * alpha_GTR1;
    }
    const float cos_theta_h = sqrtf(num / den);
    // the azimuth and the radial factor
    const float angle = diffuse ? (xi_2 * 2.0f * PI_) : (2.0f * PI_ * xi_1);
    const float radial = sqrtf(diffuse ? xi_1 : fmax_(0.0f, 1.0f - cos_theta_h * cos_theta_h));
    const float sin_a = det_sin(angle), cos_a = det_cos(angle);
    const float x = radial * cos_a;
    const float y = radial * sin_a;
    const float z = diffuse ? sqrtf(1.0f - x * x - y * y) : cos_theta_h;
    const vec3 T = toNormalHemisphere(v3(x, y, z), N);
    return diffuse ? T : reflect(-V, T);
}

// SampleBRDF (above) on the precomputed terms.
RT_HD vec3 SampleBRDF_pre(float xi_1, float xi_2, float xi_3, vec3 V, vec3 N, const MatPre& q) {
    const float rd = xi_3;
    const bool diffuse = rd <= q.p_d;
    const bool specular = !diffuse && (q.p_d < rd && rd <= q.p_d + q.p_s);
    const bool clearcoat = !diffuse && !specular && (q.p_d + q.p_s < rd);
    if (!(diffuse || specular || clearcoat)) return v3(0, 1, 0);  // NaN weights
    float num = 1.0f - xi_2, den = 1.0f + q.a2m1 * xi_2;
    if (clearcoat) {
        // det_pow(a^2, y) = exp(y * log(a^2)) with its x == 0 cases; the log comes from the table
        const float y = 1.0f - xi_2;
        const float pw = (q.g1_a2 == 0.0f) ? ((y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : u2f(0x7f800000u))) : det_exp(y * q.g1_lg);
        num = 1.0f - pw;
        den = 1.0f - q.g1_a2;
    }
    const float cos_theta_h = sqrtf(num / den);
    const float angle = diffuse ? (xi_2 * 2.0f * PI_) : (2.0f * PI_ * xi_1);
    const float radial = sqrtf(diffuse ? xi_1 : fmax_(0.0f, 1.0f - cos_theta_h * cos_theta_h));
    const float sin_a = det_sin(angle), cos_a = det_cos(angle);
    const float x = radial * cos_a;
    const float y = radial * sin_a;
    const float z = diffuse ? sqrtf(1.0f - x * x - y * y) : cos_theta_h;
    const vec3 T = toNormalHemisphere(v3(x, y, z), N);
    return diffuse ? T : reflect(-V, T);
}

// ---- E1/E2/S6: environment map, f5:612-660 -------------------------------------------------------------------------
// NEAREST + CLAMP_TO_EDGE texel index of texture2D (c5:194-197); NaN coordinates map to texel 0.
RT_HD int texel_index(float coord, int size) {
    const float f = floorf(coord * (float)size);
    return (f >= 0.0f) ? ((f <= (float)(size - 1)) ? (int)f : (size - 1)) : 0;
}
RT_HD void toSphericalCoord(vec3 v, float* u, float* w) {
    float a = det_atan2(v.z, v.x), b = det_asin(v.y);
    a = a / (2.0f * PI_);
    b = b / PI_;
    a = a + 0.5f;
    b = b + 0.5f;
    *u = a;
    *w = 1.0f - b;
}
RT_HD vec3 hdrDirection(float cx, float cy) {  // SampleHdr after the hdrCache fetch, f5:615-624
    const float y = 1.0f - cy;
    const float phi = 2.0f * PI_ * (cx - 0.5f);
    const float theta = PI_ * (y - 0.5f);
    return v3(det_cos(theta) * det_cos(phi), det_sin(theta), det_cos(theta) * det_sin(phi));
}
RT_HD float hdrPdfConvert(float cache_b, float uv_y, int hdrResolution) {  // f5:653-659
    const float theta = PI_ * (0.5f - uv_y);
    const float sin_theta = fmax_(det_sin(theta), 1e-10f);
    const float p_convert = (float)(hdrResolution * hdrResolution / 2) / (2.0f * PI_ * PI_ * sin_theta);
    return cache_b * p_convert;
}
RT_HD float misMixWeight(float a, float b) {
    const float t = a * a;
    return t / (b * b + t);
}

}  // namespace rta
