// vecmath.h -- fp64 small-vector algebra and the reverse-mode pieces the renderer needs.
//
// All shading arithmetic is fp64, like the reference (`using Real = double`,
// /root/reference/src/redner.h:46); geometry/texture storage at the boundary is fp32.  Where the
// reference mixes float literals into double expressions (e.g. luminance weights,
// src/vector.h:506-510; coordinate_system, :529-541) the float-rounded constant is part of the
// result, so the same literals (with the `f` suffix) are used here.
//
// RDR_FN comes from the exec header in use: `__device__ inline` for the gfx950 build
// (csrc/hip/exec.h).
#pragma once
#include "exec.h"   // resolved by include path: csrc/hip/exec.h for the gfx950 build
#include <math.h>
#include <stdint.h>

// Floating-point contract.  The device build runs with -ffp-contract=off because the geometric
// predicates of the edge estimators (silhouette tests, LTC-space clipping) evaluate quantities that
// are exactly zero up to rounding for edges lying in the shading point's own plane, so their sign --
// and with it which edge a sample picks -- must be rounded exactly like the reference's CPU build.
// Functions whose results only enter continuous quantities (BSDF values, adjoints) opt back in to
// fused multiply-add with RDR_CONTRACT_FAST as their first statement.
#if defined(__clang__)
#define RDR_CONTRACT_FAST _Pragma("clang fp contract(fast)")
#else
#define RDR_CONTRACT_FAST
#endif

#include "libm_exact.h"

namespace rdr {

// The seven transcendental functions of the path are glibc's, bit for bit, on the device as on the oracle's host
// (libm_exact.h): inside namespace rdr an unqualified sin(x) ... is gm::sin(x).  All call sites pass doubles, like the
// reference's Real; a float argument would have selected glibc's float routine there and is a compile error here.
#ifndef RDR_PLATFORM_LIBM    // defined for the DEFAULT product build (libredner_amd.so: the device's own libm, __graft_entry__.build_native);
                             // undefined: the glibc-exact routines of libm_exact.h (libredner_amd_exact.so, the CPU harness)
using gm::sin; using gm::cos; using gm::atan2; using gm::atan; using gm::acos; using gm::log; using gm::pow;
#endif

// One object read from DEVICE memory through a pointer that was itself loaded from memory (mesh records, texels: the tables of
// a Scene): the compiler knows no address space for such a pointer and emits flat_load, which counts against the vector-memory
// AND the LDS / scalar-memory counters -- every wait on a scalar load then also waits for it.  These tables are hipMalloc'ed
// blocks, always: read them as global memory.  (The copy is scalarised: fields that are not used are not loaded.)
template <class T> RDR_FN T load_dev(const T *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    T v;
    __builtin_memcpy(&v, (const __attribute__((address_space(1))) void *)p, sizeof(T));
    return v;
#else
    return *p;
#endif
}

struct V2 { double x, y; };
struct V3 { double x, y, z; };

RDR_FN V2 v2(double x, double y) { return V2{x, y}; }
RDR_FN V3 v3(double x, double y, double z) { return V3{x, y, z}; }
RDR_FN V3 v3(double s) { return V3{s, s, s}; }
RDR_FN V3 v3f(const float *p) { return V3{(double)p[0], (double)p[1], (double)p[2]}; }

RDR_FN V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
RDR_FN V2 operator-(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }
RDR_FN V2 operator-(V2 a) { return V2{-a.x, -a.y}; }
RDR_FN V2 operator*(V2 a, double s) { return V2{a.x * s, a.y * s}; }
RDR_FN V2 operator*(double s, V2 a) { return V2{s * a.x, s * a.y}; }
RDR_FN V2 operator*(V2 a, V2 b) { return V2{a.x * b.x, a.y * b.y}; }
// vector / scalar multiplies by the reciprocal, like the reference (src/vector.h:287-299): the
// 1-ulp difference from a true division decides ties in the edge-hierarchy build.
RDR_FN V2 operator/(V2 a, double s) { double inv = 1.f / s; return V2{a.x * inv, a.y * inv}; }
RDR_FN V2 &operator+=(V2 &a, V2 b) { a.x += b.x; a.y += b.y; return a; }
RDR_FN V2 &operator-=(V2 &a, V2 b) { a.x -= b.x; a.y -= b.y; return a; }
RDR_FN double dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
RDR_FN double sum(V2 a) { return a.x + a.y; }
RDR_FN double len_sq(V2 a) { return a.x * a.x + a.y * a.y; }
RDR_FN double len(V2 a) { return sqrt(len_sq(a)); }
RDR_FN V2 normalize(V2 a) { return a / len(a); }

RDR_FN V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RDR_FN V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RDR_FN V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RDR_FN V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
RDR_FN V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
RDR_FN V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RDR_FN V3 operator/(V3 a, double s) { double inv = 1.f / s; return V3{a.x * inv, a.y * inv, a.z * inv}; }
RDR_FN V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
RDR_FN V3 operator+(V3 a, double s) { return V3{a.x + s, a.y + s, a.z + s}; }
RDR_FN V3 operator-(V3 a, double s) { return V3{a.x - s, a.y - s, a.z - s}; }
RDR_FN V3 operator-(double s, V3 a) { return V3{s - a.x, s - a.y, s - a.z}; }
RDR_FN V3 &operator+=(V3 &a, V3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
RDR_FN V3 &operator-=(V3 &a, V3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
RDR_FN V3 &operator*=(V3 &a, double s) { a.x *= s; a.y *= s; a.z *= s; return a; }
RDR_FN double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RDR_FN double sum(V3 a) { return a.x + a.y + a.z; }
RDR_FN V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
RDR_FN double len_sq(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RDR_FN double len(V3 a) { return sqrt(len_sq(a)); }
RDR_FN bool all_zero(V3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
RDR_FN double comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
RDR_FN V3 vmax0(V3 a) { return V3{a.x > 0 ? a.x : 0.0, a.y > 0 ? a.y : 0.0, a.z > 0 ? a.z : 0.0}; }
RDR_FN double dmax(double a, double b) { return a > b ? a : b; }
RDR_FN double dmin(double a, double b) { return a < b ? a : b; }
RDR_FN int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
RDR_FN double sq(double x) { return x * x; }
RDR_FN void accum3(double *p, V3 v) { accum_triple(p, v.x, v.y, v.z); }      // one search for the lanes that share the triple (exec.h)
RDR_FN void accum3_plain(double *p, V3 v) { accum_plain(p, v.x); accum_plain(p + 1, v.y); accum_plain(p + 2, v.z); }

// normalize(0) == 0 (src/vector.h:448-457)
RDR_FN V3 normalize(V3 a) {
    double l = len(a);
    if (l <= 0) return V3{0, 0, 0};
    return a / l;
}

// luminance with float-rounded weights (src/vector.h:506-510)
RDR_FN double luminance(V3 c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; }

// ---- reverse-mode helpers.  "bar" arguments are incoming adjoints. ---------------------------

// adjoint of l = |a|^2
RDR_FN V3 adj_len_sq(V3 a, double l_bar) { return (2 * l_bar) * a; }
RDR_FN V2 adj_len_sq(V2 a, double l_bar) { return (2 * l_bar) * a; }
// adjoint of l = |a|   (0.5f literal as in src/vector.h:395-401)
RDR_FN V3 adj_len(V3 a, double l_bar) {
    double l = sqrt(len_sq(a));
    return adj_len_sq(a, 0.5f * l_bar / l);
}
RDR_FN V2 adj_len(V2 a, double l_bar) {
    double l = sqrt(len_sq(a));
    return adj_len_sq(a, 0.5f * l_bar / l);
}
// adjoint of n = normalize(a)  (src/vector.h:459-472)
RDR_FN V3 adj_normalize(V3 a, V3 n_bar) {
    double l = len(a);
    if (l <= 0) return V3{0, 0, 0};
    V3 n = a / l;
    V3 a_bar = n_bar / l;
    double l_bar = -dot(n_bar, n) / l;
    a_bar += adj_len(a, l_bar);
    return a_bar;
}
// adjoint of c = cross(a, b)
RDR_FN void adj_cross(V3 a, V3 b, V3 c_bar, V3 &a_bar, V3 &b_bar) {
    a_bar += cross(b, c_bar);
    b_bar += cross(c_bar, a);
}

// Orthonormal basis around n (Frisvad form used by the reference, src/vector.h:527-541).
RDR_FN void onb(V3 n, V3 &x, V3 &y) {
    if (n.z < -1.f + 1e-6f) {
        x = V3{0, -1, 0};
        y = V3{-1, 0, 0};
    } else {
        double a = 1.f / (1.f + n.z);
        double b = -n.x * n.y * a;
        x = V3{1.f - n.x * n.x * a, b, -n.x};
        y = V3{b, 1.f - n.y * n.y * a, -n.y};
    }
}
RDR_FN void adj_onb(V3 n, V3 x_bar, V3 y_bar, V3 &n_bar) {
    if (n.z < -1.f + 1e-6f) return;
    double a = 1.f / (1.f + n.z);
    // x = (1 - nx^2 a, b, -nx),  y = (b, 1 - ny^2 a, -ny),  b = -nx ny a
    double a_bar = -(n.x * n.x) * x_bar.x - (n.y * n.y) * y_bar.y;
    double b_bar = x_bar.y + y_bar.x;
    n_bar.x -= 2.f * n.x * x_bar.x * a;
    n_bar.x -= x_bar.z;
    n_bar.y -= 2.f * y_bar.y * n.y * a;
    n_bar.y -= y_bar.z;
    n_bar.x -= b_bar * n.y * a;
    n_bar.y -= b_bar * n.x * a;
    a_bar -= b_bar * n.x * n.y;
    n_bar.z -= a_bar * a / (1 + n.z);
}

// Frame = (x, y, n)
struct Frame { V3 x, y, n; };
RDR_FN Frame frame_from_normal(V3 n) { Frame f; f.n = n; onb(n, f.x, f.y); return f; }
RDR_FN Frame frame_zero() { return Frame{V3{0, 0, 0}, V3{0, 0, 0}, V3{0, 0, 0}}; }
RDR_FN V3 to_local(const Frame &f, V3 v) { return V3{dot(v, f.x), dot(v, f.y), dot(v, f.n)}; }
RDR_FN V3 to_world(const Frame &f, V3 v) { return f.x * v.x + f.y * v.y + f.n * v.z; }
RDR_FN void adj_to_world(const Frame &f, V3 v, V3 w_bar, Frame &f_bar, V3 &v_bar) {
    f_bar.x += w_bar * v.x; f_bar.y += w_bar * v.y; f_bar.n += w_bar * v.z;
    v_bar.x += sum(w_bar * f.x); v_bar.y += sum(w_bar * f.y); v_bar.z += sum(w_bar * f.n);
}

// Row-major small matrices (src/matrix.h:20-28: data[r][c] = arr[3r+c]).
struct M3 { double m[3][3]; };
struct M4 { double m[4][4]; };
RDR_FN M3 m3_zero() { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = 0; return r; }
RDR_FN M4 m4_zero() { M4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = 0; return r; }
RDR_FN V3 mul(const M3 &a, V3 v) {
    return V3{a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
              a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
              a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
RDR_FN V3 mul_t(const M3 &a, V3 v) {   // transpose(a) * v  ==  v * a
    return V3{a.m[0][0] * v.x + a.m[1][0] * v.y + a.m[2][0] * v.z,
              a.m[0][1] * v.x + a.m[1][1] * v.y + a.m[2][1] * v.z,
              a.m[0][2] * v.x + a.m[1][2] * v.y + a.m[2][2] * v.z};
}
// Homogeneous point transform with the 1.f / w of src/transform.h:75-86.
RDR_FN V3 xfm_point(const M4 &a, V3 p) {
    double tx = a.m[0][0] * p.x + a.m[0][1] * p.y + a.m[0][2] * p.z + a.m[0][3];
    double ty = a.m[1][0] * p.x + a.m[1][1] * p.y + a.m[1][2] * p.z + a.m[1][3];
    double tz = a.m[2][0] * p.x + a.m[2][1] * p.y + a.m[2][2] * p.z + a.m[2][3];
    double tw = a.m[3][0] * p.x + a.m[3][1] * p.y + a.m[3][2] * p.z + a.m[3][3];
    double iw = 1.f / tw;
    return V3{tx, ty, tz} * iw;
}
RDR_FN V3 xfm_vector(const M4 &a, V3 v) {
    return V3{a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
              a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
              a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
RDR_FN void adj_xfm_point(const M4 &a, V3 p, V3 o_bar, M4 &a_bar, V3 &p_bar) {
    double t[4];
    for (int r = 0; r < 4; ++r) t[r] = a.m[r][0] * p.x + a.m[r][1] * p.y + a.m[r][2] * p.z + a.m[r][3];
    double iw = 1.f / t[3];
    double tb[4];
    tb[0] = o_bar.x * iw; tb[1] = o_bar.y * iw; tb[2] = o_bar.z * iw;
    double iw_bar = o_bar.x * t[0] + o_bar.y * t[1] + o_bar.z * t[2];
    tb[3] = -iw_bar * iw * iw;
    for (int r = 0; r < 4; ++r) {
        a_bar.m[r][0] += tb[r] * p.x; a_bar.m[r][1] += tb[r] * p.y;
        a_bar.m[r][2] += tb[r] * p.z; a_bar.m[r][3] += tb[r];
    }
    p_bar.x += tb[0] * a.m[0][0] + tb[1] * a.m[1][0] + tb[2] * a.m[2][0] + tb[3] * a.m[3][0];
    p_bar.y += tb[0] * a.m[0][1] + tb[1] * a.m[1][1] + tb[2] * a.m[2][1] + tb[3] * a.m[3][1];
    p_bar.z += tb[0] * a.m[0][2] + tb[1] * a.m[1][2] + tb[2] * a.m[2][2] + tb[3] * a.m[3][2];
}
RDR_FN void adj_xfm_vector(const M4 &a, V3 v, V3 o_bar, M4 &a_bar, V3 &v_bar) {
    double ob[3] = {o_bar.x, o_bar.y, o_bar.z};
    for (int r = 0; r < 3; ++r) {
        a_bar.m[r][0] += ob[r] * v.x; a_bar.m[r][1] += ob[r] * v.y; a_bar.m[r][2] += ob[r] * v.z;
    }
    v_bar.x += ob[0] * a.m[0][0] + ob[1] * a.m[1][0] + ob[2] * a.m[2][0];
    v_bar.y += ob[0] * a.m[0][1] + ob[1] * a.m[1][1] + ob[2] * a.m[2][1];
    v_bar.z += ob[0] * a.m[0][2] + ob[1] * a.m[1][2] + ob[2] * a.m[2][2];
}

// look-at: columns = (right, new_up, d, pos)  (src/transform.h:9-29) and its adjoint (:31-73)
inline M4 look_at(V3 pos, V3 look, V3 up) {
    V3 d = normalize(look - pos);
    V3 right = normalize(cross(d, normalize(up)));
    V3 nup = normalize(cross(right, d));
    M4 r = m4_zero();
    r.m[0][0] = right.x; r.m[0][1] = nup.x; r.m[0][2] = d.x; r.m[0][3] = pos.x;
    r.m[1][0] = right.y; r.m[1][1] = nup.y; r.m[1][2] = d.y; r.m[1][3] = pos.y;
    r.m[2][0] = right.z; r.m[2][1] = nup.z; r.m[2][2] = d.z; r.m[2][3] = pos.z;
    r.m[3][3] = 1;
    return r;
}
RDR_FN void adj_look_at(V3 pos, V3 look, V3 up, const M4 &m_bar, V3 &pos_bar, V3 &look_bar, V3 &up_bar) {
    V3 lp = look - pos;
    V3 d = normalize(lp);
    V3 nu = normalize(up);
    V3 cdu = cross(d, nu);
    V3 right = normalize(cdu);
    V3 crd = cross(right, d);
    V3 right_bar = V3{m_bar.m[0][0], m_bar.m[1][0], m_bar.m[2][0]};
    V3 nup_bar = V3{m_bar.m[0][1], m_bar.m[1][1], m_bar.m[2][1]};
    V3 d_bar = V3{m_bar.m[0][2], m_bar.m[1][2], m_bar.m[2][2]};
    pos_bar += V3{m_bar.m[0][3], m_bar.m[1][3], m_bar.m[2][3]};
    V3 crd_bar = adj_normalize(crd, nup_bar);
    adj_cross(right, d, crd_bar, right_bar, d_bar);
    V3 cdu_bar = adj_normalize(cdu, right_bar);
    V3 nu_bar = V3{0, 0, 0};
    adj_cross(d, nu, cdu_bar, d_bar, nu_bar);
    up_bar += adj_normalize(up, nu_bar);
    V3 lp_bar = adj_normalize(lp, d_bar);
    look_bar += lp_bar;
    pos_bar -= lp_bar;
}

// General 4x4 inverse (host only; used once per camera).
inline M4 inverse(const M4 &a) {
    double s[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { s[i][j] = a.m[i][j]; s[i][j + 4] = (i == j); }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (fabs(s[r][c]) > fabs(s[piv][c])) piv = r;
        if (piv != c) for (int j = 0; j < 8; ++j) { double t = s[c][j]; s[c][j] = s[piv][j]; s[piv][j] = t; }
        double inv = 1.0 / s[c][c];
        for (int j = 0; j < 8; ++j) s[c][j] *= inv;
        for (int r = 0; r < 4; ++r) if (r != c) {
            double f = s[r][c];
            if (f != 0) for (int j = 0; j < 8; ++j) s[r][j] -= f * s[c][j];
        }
    }
    M4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = s[i][j + 4];
    return r;
}

} // namespace rdr
