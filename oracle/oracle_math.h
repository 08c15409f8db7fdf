// ORACLE (test infrastructure, not product code): the oracle's own copy of the build's deterministic fp32 transcendentals.
//
// GLSL leaves sin/cos/atan/asin/log/pow precision to the driver, so the reference pins no particular rounding for them.
// This build fixes one: the sequences below use only IEEE fp32 + - * / sqrt and integer ops (no FMA: compile with
// -ffp-contract=off), so the HIP kernels and the CPU oracle (oracle/oracle_math.h holds the oracle's own copy) produce
// bit-identical images.  Polynomials are the classic single-precision Cephes ones (S. Moshier), ~1-2 ulp.
#pragma once
#include <math.h>
#define RT_HD inline

namespace orc {

RT_HD unsigned f2u(float f) { return __builtin_bit_cast(unsigned, f); }
RT_HD float u2f(unsigned u) { return __builtin_bit_cast(float, u); }

constexpr float kPiF = 3.14159265358979323846f;
constexpr float kPiO2F = 1.5707963267948966192f;
constexpr float kPiO4F = 0.7853981633974483096f;

// shared octant reduction: returns the reduced argument, octant (0..7, even) in *oct
RT_HD float det_reduce_pio4(float ax, int* oct) {
    int j = (int)(1.27323954473516f * ax);  // 4/pi
    float y = (float)j;
    if (j & 1) {
        j += 1;
        y += 1.0f;
    }
    *oct = j & 7;
    return ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
}
RT_HD float det_sin_poly(float x, float z) {
    return ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x + x;
}
RT_HD float det_cos_poly(float z) {
    float y = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}

RT_HD float det_sin(float x) {
    float sign = 1.0f;
    if (x < 0.0f) {
        x = -x;
        sign = -1.0f;
    }
    if (!(x <= 8192.0f)) return 0.0f;  // total loss of precision (and NaN)
    int j;
    const float r = det_reduce_pio4(x, &j);
    if (j > 3) {
        sign = -sign;
        j -= 4;
    }
    const float z = r * r;
    const float y = (j == 1 || j == 2) ? det_cos_poly(z) : det_sin_poly(r, z);
    return (sign < 0.0f) ? -y : y;
}

RT_HD float det_cos(float x) {
    if (x < 0.0f) x = -x;
    if (!(x <= 8192.0f)) return 0.0f;
    int j;
    const float r = det_reduce_pio4(x, &j);
    float sign = 1.0f;
    if (j > 3) {
        j -= 4;
        sign = -sign;
    }
    if (j > 1) sign = -sign;
    const float z = r * r;
    const float y = (j == 1 || j == 2) ? det_sin_poly(r, z) : det_cos_poly(z);
    return (sign < 0.0f) ? -y : y;
}

RT_HD float det_atan(float xx) {
    float sign = 1.0f, x = xx;
    if (xx < 0.0f) {
        sign = -1.0f;
        x = -xx;
    }
    float y;
    if (x > 2.414213562373095f) {  // tan(3pi/8)
        y = kPiO2F;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {  // tan(pi/8)
        y = kPiO4F;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    const float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * x + x);
    return (sign < 0.0f) ? -y : y;
}

// GLSL atan(y, x)
RT_HD float det_atan2(float y, float x) {
    if (x == 0.0f) return (y > 0.0f) ? kPiO2F : ((y < 0.0f) ? -kPiO2F : 0.0f);
    const float z = det_atan(y / x);
    if (x < 0.0f) return (y < 0.0f) ? (z - kPiF) : (z + kPiF);
    return z;
}

// asin with the argument clamped to [-1, 1] (normalize() can overshoot by an ulp)
RT_HD float det_asin(float xx) {
    float sign = 1.0f, a = xx;
    if (xx < 0.0f) {
        sign = -1.0f;
        a = -xx;
    }
    if (!(a <= 1.0f)) a = 1.0f;
    if (a < 1.0e-4f) return (sign < 0.0f) ? -a : a;
    float x, z;
    const bool flag = a > 0.5f;
    if (flag) {
        z = 0.5f * (1.0f - a);
        x = sqrtf(z);
    } else {
        x = a;
        z = x * x;
    }
    z = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z + 7.4953002686E-2f) * z + 1.6666752422E-1f) * z * x + x;
    if (flag) {
        z = z + z;
        z = kPiO2F - z;
    }
    return (sign < 0.0f) ? -z : z;
}

// natural log; x <= 0 -> -inf / NaN like logf
RT_HD float det_log(float x) {
    if (!(x > 0.0f)) return (x == 0.0f) ? u2f(0xff800000u) : u2f(0x7fc00000u);
    if (f2u(x) >= 0x7f800000u) return x;  // +inf
    int e = 0;
    if (f2u(x) < 0x00800000u) {  // subnormal: scale by 2^24 (exact)
        x = x * 16777216.0f;
        e = -24;
    }
    const unsigned u = f2u(x);
    e += (int)(u >> 23) - 126;                          // frexp: x = m * 2^e, m in [0.5, 1)
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float y = ((((((((7.0376836292E-2f * m - 1.1514610310E-1f) * m + 1.1676998740E-1f) * m - 1.2420140846E-1f) * m +
                   1.4249322787E-1f) * m - 1.6668057665E-1f) * m + 2.0000714765E-1f) * m - 2.4999993993E-1f) * m +
               3.3333331174E-1f) * m * z;
    const float fe = (float)e;
    if (e != 0) y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = m + y;
    if (e != 0) r = r + 0.693359375f * fe;
    return r;
}

RT_HD float det_exp(float x) {
    if (!(x == x)) return x;
    if (x > 88.72283905206835f) return u2f(0x7f800000u);
    if (x < -103.278929903431851103f) return 0.0f;
    const float n = floorf(1.44269504088896341f * x + 0.5f);
    x = x - n * 0.693359375f;
    x = x - n * -2.12194440e-4f;
    const float z = x * x;
    float r = (((((1.9875691500E-4f * x + 1.3981999507E-3f) * x + 8.3334519073E-3f) * x + 4.1665795894E-2f) * x +
                1.6666665459E-1f) * x + 5.0000001201E-1f) * z + x + 1.0f;
    // ldexp in two exact steps so that subnormal results round once
    int k = (int)n;
    if (k > 127) {
        r = r * u2f((unsigned)(127 + 127) << 23);
        k -= 127;
    } else if (k < -126) {
        r = r * u2f((unsigned)(-100 + 127) << 23);
        k += 100;
    }
    return r * u2f((unsigned)(k + 127) << 23);
}

// GLSL pow(x, y) = exp(y*log(x)); undefined for x < 0 (NaN here), pow(0, y>0) = 0
RT_HD float det_pow(float x, float y) {
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : u2f(0x7f800000u));
    return det_exp(y * det_log(x));
}

}  // namespace orc
