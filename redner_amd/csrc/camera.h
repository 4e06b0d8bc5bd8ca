// camera.h -- camera models (perspective, orthographic, fisheye, panorama) with optional
// Brown-Conrady lens distortion: primary rays, projection of world segments to the screen, and
// their adjoints.
//
// Behavioural spec: sample_primary src/camera.h:121-197, d_sample_primary_ray :199-494,
// camera_to_screen :508-559, project :561-591, d_camera_to_screen :595-729, d_project :731-830,
// screen_to_camera :832-897, d_screen_to_camera :899-1047, in_screen :1049-1067; distort /
// inverse_distort and adjoints src/camera_distortion.h:17-259; primary ray + finite-difference
// ray differential src/camera.cpp:8-43.  Reference quirks that are reproduced are tagged [quirk].
#pragma once
#include "surface.h"

namespace rdr {

enum { kCamPerspective = 0, kCamOrthographic = 1, kCamFisheye = 2, kCamPanorama = 3 };

RDR_FN double aspect_of(const CameraD &cam) { return double(cam.width) / double(cam.height); }

RDR_FN Ray make_ray(V3 org, V3 dir) { return Ray{org, dir, double(1e-3f), INFINITY}; }

// ---- lens distortion -------------------------------------------------------------------------------
struct DistortTerms { double x, y, r, r2, r4, r6, num, den, rr; };
RDR_FN DistortTerms distort_terms(const DistortD &d, V2 pos) {
    DistortTerms t;
    t.x = 2.f * (pos.x - 0.5f); t.y = 2.f * (pos.y - 0.5f);
    t.r = sqrt(t.x * t.x + t.y * t.y);
    t.r2 = t.r * t.r; t.r4 = t.r2 * t.r2; t.r6 = t.r4 * t.r2;
    t.num = 1 + d.k[0] * t.r2 + d.k[1] * t.r4 + d.k[2] * t.r6;
    t.den = 1 + d.k[3] * t.r2 + d.k[4] * t.r4 + d.k[5] * t.r6;
    t.rr = t.num / t.den;
    return t;
}
// Distorted position; optionally the forward-mode Jacobian rows d(out.x)/d(pos), d(out.y)/d(pos).
RDR_FN V2 distort(const DistortD &d, V2 pos, V2 *dx_dpos = nullptr, V2 *dy_dpos = nullptr) {
    if (!d.defined) return pos;
    DistortTerms t = distort_terms(d, pos);
    double x = t.x, y = t.y;
    double xx = x * t.rr + 2 * d.p[0] * x * y + d.p[1] * (t.r2 + 2 * x * x);
    double yy = y * t.rr + d.p[0] * (t.r2 + 2 * y * y) + 2 * d.p[1] * x * y;
    if (dx_dpos && dy_dpos) {
        V2 dx = 2 * v2(1, 0), dy = 2 * v2(0, 1);
        V2 dr = (dx * x + dy * y) / t.r;
        V2 dr2 = 2 * t.r * dr;
        V2 dr4 = 2 * t.r2 * dr2;
        V2 dr6 = t.r4 * dr2 + dr4 * t.r2;
        V2 dnum = (d.k[0] * dr2 + d.k[1] * dr4 + d.k[2] * dr6);
        V2 dden = (d.k[3] * dr2 + d.k[4] * dr4 + d.k[5] * dr6);
        V2 drr = (dnum * t.den - t.num * dden) / (t.den * t.den);
        V2 dxx = dx * t.rr + x * drr + 2 * d.p[0] * (dx * y + x * dy) + d.p[1] * (dr2 + 4 * dx * x);
        V2 dyy = dy * t.rr + y * drr + d.p[0] * (dr2 + 4 * dy * y) + 2 * d.p[1] * (dx * y + x * dy);
        *dx_dpos = dxx / 2;
        *dy_dpos = dyy / 2;
    }
    return v2((xx + 1) / 2, (yy + 1) / 2);
}
// Adjoint of distort(): pos_bar += ..., parameter gradient into g_dist[8] (k0..k5, p0, p1) when non-null.
RDR_FN void adj_distort(const DistortD &d, V2 pos, V2 out_bar, double *g_dist, V2 &pos_bar) {
    if (!d.defined) { pos_bar = out_bar; return; }
    DistortTerms t = distort_terms(d, pos);
    double x = t.x, y = t.y, r = t.r, r2 = t.r2, r4 = t.r4, r6 = t.r6, rr = t.rr;
    double k_bar[6] = {0, 0, 0, 0, 0, 0}, p_bar[2] = {0, 0};
    double xx_bar = out_bar.x / 2, yy_bar = out_bar.y / 2;
    double x_bar = xx_bar * (rr + 2 * d.p[0] * y + 4 * d.p[1] * x);
    double rr_bar = xx_bar * x;
    double y_bar = xx_bar * 2 * d.p[0] * x;
    p_bar[0] += xx_bar * 2 * x * y;
    p_bar[1] += xx_bar * (r2 + 2 * x * x);
    double r2_bar = xx_bar * d.p[1];
    y_bar += yy_bar * (rr + 4 * d.p[0] * y + 2 * d.p[1] * x);
    rr_bar += yy_bar * y;
    p_bar[0] += yy_bar * (r2 + 2 * y * y);
    r2_bar += yy_bar * d.p[0];
    p_bar[1] += yy_bar * 2 * x * y;
    x_bar += yy_bar * 2 * d.p[1] * y;
    double num_bar = rr_bar / t.den;
    double den_bar = -rr_bar * rr / t.den;
    k_bar[0] += num_bar * r2; r2_bar += num_bar * d.k[0];
    k_bar[1] += num_bar * r4; double r4_bar = num_bar * d.k[1];
    k_bar[2] += num_bar * r6; double r6_bar = num_bar * d.k[2];
    k_bar[3] += den_bar * r2; r2_bar += den_bar * d.k[3];
    k_bar[4] += den_bar * r4; r4_bar += den_bar * d.k[4];
    k_bar[5] += den_bar * r6; r6_bar += den_bar * d.k[5];
    r4_bar += r6_bar * r2;
    r2_bar += r6_bar * r2;            // [quirk] r2, where the derivative of r4*r2 w.r.t. r2 is r4 (camera_distortion.h:150)
    r2_bar += 2 * r4_bar * r2;
    double r_bar = 2 * r2_bar * r;
    x_bar += r_bar * x / r;
    y_bar += r_bar * y / r;
    pos_bar.x += x_bar * 2;
    pos_bar.y += y_bar * 2;
    if (g_dist) {
        _Pragma("unroll") for (int i = 0; i < 6; ++i) accum(g_dist + i, k_bar[i]);
        accum(g_dist + 6, p_bar[0]);
        accum(g_dist + 7, p_bar[1]);
    }
}
// Gauss-Newton inversion of distort().
RDR_FN V2 inverse_distort(const DistortD &d, V2 pos) {
    if (!d.defined) return pos;
    V2 result = pos;
    double err = 0;
    int iter = 0;
    do {
        V2 jx = v2(0, 0), jy = v2(0, 0);
        V2 next = distort(d, result, &jx, &jy);
        V2 residual = next - pos;
        err = fabs(residual.x) + fabs(residual.y);
        double det = jx.x * jy.y - jx.y * jy.x;
        double inv_det = 1 / det;
        V2 inv0 = inv_det * v2(jy.y, -jx.y), inv1 = inv_det * v2(-jy.x, jx.x);
        result = result - v2(dot(inv0, residual), dot(inv1, residual));
    } while (err > 1e-3 && iter++ < 1000);
    return result;
}
// Adjoint through the implicit function theorem (camera_distortion.h:208-259).
RDR_FN void adj_inverse_distort(const DistortD &d, V2 pos, V2 out_bar, double *g_dist, V2 &pos_bar) {
    if (!d.defined) { pos_bar = out_bar; return; }
    V2 result = inverse_distort(d, pos);
    V2 fx = v2(0, 0), fy = v2(0, 0);
    distort(d, result, &fx, &fy);
    double det = fx.x * fy.y - fx.y * fy.x;
    double inv_det = 1 / det;
    V2 it0 = inv_det * v2(fy.y, -fy.x), it1 = inv_det * v2(-fx.y, fx.x);
    V2 result_bar = -v2(dot(it0, out_bar), dot(it1, out_bar));
    V2 unused = v2(0, 0);
    if (g_dist) adj_distort(d, result, result_bar, g_dist, unused);
    pos_bar -= result_bar;
}

// ---- primary rays ----------------------------------------------------------------------------------
RDR_FN V3 fisheye_dir(double x, double y) {
    double r = sqrt(x * x + y * y);
    double phi = atan2(y, x);
    double theta = r * double(M_PI / 2);
    double sp = sin(phi), cp = cos(phi), st = sin(theta), ct = cos(theta);
    return V3{-cp * st, -sp * st, ct};
}
RDR_FN V3 panorama_dir(V2 s) {
    double theta = double(M_PI) * s.y, phi = double(2 * M_PI) * s.x;
    double sp = sin(phi), cp = cos(phi), st = sin(theta), ct = cos(theta);
    return V3{cp * st, ct, sp * st};
}

RDR_FN Ray primary_ray(const CameraD &cam, V2 screen_in) {
    V2 screen = inverse_distort(cam.distortion, screen_in);
    switch (cam.kind) {
        case kCamOrthographic: {
            double ar = aspect_of(cam);
            V3 pt = V3{(screen.x - 0.5f) * 2.f, (screen.y - 0.5f) * (-2.f) / ar, 0.0};
            V3 org = xfm_point(cam.cam_to_world, mul(cam.intrinsic_mat_inv, pt));
            V3 dir = normalize(xfm_vector(cam.cam_to_world, V3{0, 0, 1}));
            return make_ray(org, dir);
        }
        case kCamFisheye: {
            V3 org = xfm_point(cam.cam_to_world, v3(0));
            double x = 2.f * (screen.x - 0.5f), y = 2.f * (screen.y - 0.5f);
            if (x * x + y * y > 1.f) return make_ray(v3(0), v3(0));
            V3 dw = normalize(xfm_vector(cam.cam_to_world, fisheye_dir(x, y)));
            return make_ray(org, dw);
        }
        case kCamPanorama: {
            V3 org = xfm_point(cam.cam_to_world, v3(0));
            V3 dw = normalize(xfm_vector(cam.cam_to_world, panorama_dir(screen)));
            return make_ray(org, dw);
        }
        default: {
            V3 org = xfm_point(cam.cam_to_world, v3(0));
            double ar = aspect_of(cam);
            V3 pt = V3{(screen.x - 0.5f) * 2.f, (screen.y - 0.5f) * (-2.f) / ar, 1.0};
            V3 dl = normalize(mul(cam.intrinsic_mat_inv, pt));
            V3 dw = normalize(xfm_vector(cam.cam_to_world, dl));
            return make_ray(org, dw);
        }
    }
}

// Screen position of sample `s` in pixel `pixel` of the viewport.
RDR_FN V2 pixel_to_screen(const CameraD &cam, int pixel, V2 s) {
    int vw = cam.vp_x1 - cam.vp_x0;
    int row = pixel / vw;
    if (cam.batch_rows > 0) row %= cam.batch_rows;          // a lane of a sample batch: the pixel of its own sample
    int px = pixel % vw + cam.vp_x0, py = row + cam.vp_y0;
    return v2((px + s.x) / double(cam.width), (py + s.y) / double(cam.height));
}

// Primary ray and its finite-difference differential (delta = 1e-3 screen units).
RDR_FN Ray primary_ray_with_diff(const CameraD &cam, V2 screen, RayDiff &rd) {
    Ray r = primary_ray(cam, screen);
    double delta = 1e-3;
    Ray rx = primary_ray(cam, screen + v2(delta, 0)), ry = primary_ray(cam, screen + v2(0, delta));
    double sx = 0.5 / cam.width, sy = 0.5 / cam.height;
    rd.org_dx = sx * (rx.org - r.org) / delta;
    rd.org_dy = sy * (ry.org - r.org) / delta;
    rd.dir_dx = sx * (rx.dir - r.dir) / delta;
    rd.dir_dy = sy * (ry.dir - r.dir) / delta;
    return r;
}

// Scatter a cam_to_world adjoint into the camera parameterisation in use.
RDR_FN void scatter_cam_to_world(const CameraD &cam, const M4 &c2w_bar, const GCamera &g) {
    if (cam.use_look_at) {
        V3 pb = v3(0), lb = v3(0), ub = v3(0);
        adj_look_at(cam.position, cam.look, cam.up, c2w_bar, pb, lb, ub);
        if (g.position) accum3(g.position, pb);
        if (g.look) accum3(g.look, lb);
        if (g.up) accum3(g.up, ub);
    } else if (g.cam_to_world) {
        _Pragma("unroll") for (int r = 0; r < 4; ++r) { _Pragma("unroll") for (int c = 0; c < 4; ++c) accum(g.cam_to_world + 4 * r + c, c2w_bar.m[r][c]); }
    }
}
RDR_FN void accum_outer3(double *g, V3 a, V3 b, int rows) {          // g[r][c] += a[r] * b[c]
    if (!g) return;
    double av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
    _Pragma("unroll") for (int r = 0; r < 3; ++r) { if (r < rows) accum_triple(g + 3 * r, av[r] * bv[0], av[r] * bv[1], av[r] * bv[2]); }      // a row: one search
}
// Tail shared by every camera model: undo the lens distortion on the way back to the screen position.
RDR_FN void adj_screen_tail(const CameraD &cam, V2 screen, V2 distorted_bar, const GCamera &g, bool want, V2 &screen_bar) {
    V2 sb = v2(0, 0);
    adj_inverse_distort(cam.distortion, screen, distorted_bar, g.distortion, sb);
    if (want) { screen_bar.x += sb.x; screen_bar.y += sb.y; }
}

// Adjoint of primary_ray(): pushes ray_bar into the camera gradient; optionally returns the
// screen-position adjoint.
// (want_screen_bar + reference instead of a nullable pointer: a pointer that may be null keeps the target in memory)
RDR_FN void adj_primary_ray(const CameraD &cam, V2 screen_in, const DRay &ray_bar, const GCamera &g, bool want_screen_bar,
                            V2 &screen_bar) {
    V2 screen = inverse_distort(cam.distortion, screen_in);
    const bool want_screen = cam.distortion.defined || want_screen_bar;
    double ar = aspect_of(cam);
    M4 c2w_bar = m4_zero();
    switch (cam.kind) {
        case kCamOrthographic: {
            V3 pt = V3{(screen.x - 0.5f) * 2.f, (screen.y - 0.5f) * (-2.f) / ar, 1.0};   // [quirk] z = 1 here, 0 in the forward
            V3 lorg = mul(cam.intrinsic_mat_inv, pt);
            V3 dir = xfm_vector(cam.cam_to_world, V3{0, 0, 1});
            V3 dir_bar = adj_normalize(dir, ray_bar.dir);
            V3 unused = v3(0);
            adj_xfm_vector(cam.cam_to_world, V3{0, 0, 1}, dir_bar, c2w_bar, unused);
            V3 lorg_bar = v3(0);
            adj_xfm_point(cam.cam_to_world, lorg, ray_bar.org, c2w_bar, lorg_bar);
            accum_outer3(g.intrinsic_mat_inv, lorg_bar, pt, 3);
            scatter_cam_to_world(cam, c2w_bar, g);
            if (want_screen) {
                V3 pt_bar = mul_t(cam.intrinsic_mat_inv, lorg_bar);
                adj_screen_tail(cam, screen_in, v2(pt_bar.x * 2, pt_bar.y * (-2 / ar)), g, want_screen_bar, screen_bar);
            }
        } break;
        case kCamFisheye: {
            double x = 2.f * (screen.x - 0.5f), y = 2.f * (screen.y - 0.5f);
            if (x * x + y * y > 1.f) return;
            double r = sqrt(x * x + y * y);
            double phi = atan2(y, x);
            double theta = r * double(M_PI) / 2.f;
            double sp = sin(phi), cp = cos(phi), st = sin(theta), ct = cos(theta);
            V3 dir = V3{-cp * st, -sp * st, ct};
            V3 dw = xfm_vector(cam.cam_to_world, dir);
            V3 dw_bar = adj_normalize(dw, ray_bar.dir);
            V3 dir_bar = v3(0);
            adj_xfm_vector(cam.cam_to_world, dir, dw_bar, c2w_bar, dir_bar);
            V3 o_bar = v3(0);
            adj_xfm_point(cam.cam_to_world, v3(0), ray_bar.org, c2w_bar, o_bar);
            scatter_cam_to_world(cam, c2w_bar, g);
            if (want_screen) {
                double cp_bar = dir_bar.x * (-st), sp_bar = dir_bar.y * (-st);
                double st_bar = dir_bar.x * (-cp) + dir_bar.y * (-sp), ct_bar = dir_bar.z;
                double phi_bar = cp_bar * (-sp) + sp_bar * cp;
                double theta_bar = ct_bar * (-st) + st_bar * ct;
                double r_bar = theta_bar * (double(M_PI) / 2.f);
                double x_bar = phi_bar * (-y / (x * x + y * y));
                double y_bar = phi_bar * (x / (x * x + y * y));
                x_bar += (r_bar * (x / r));
                y_bar += (r_bar * (y / r));
                adj_screen_tail(cam, screen_in, v2(2 * x_bar, 2 * y_bar), g, want_screen_bar, screen_bar);
            }
        } break;
        case kCamPanorama: {
            double theta = double(M_PI) * screen.y, phi = double(2 * M_PI) * screen.x;
            double sp = sin(phi), cp = cos(phi), st = sin(theta), ct = cos(theta);
            V3 dir = V3{cp * st, ct, sp * st};
            V3 dw = xfm_vector(cam.cam_to_world, dir);
            V3 dw_bar = adj_normalize(dw, ray_bar.dir);
            V3 dir_bar = v3(0);
            adj_xfm_vector(cam.cam_to_world, dir, dw_bar, c2w_bar, dir_bar);
            V3 o_bar = v3(0);
            adj_xfm_point(cam.cam_to_world, v3(0), ray_bar.org, c2w_bar, o_bar);
            scatter_cam_to_world(cam, c2w_bar, g);
            if (want_screen) {
                double cp_bar = dir_bar.x * st, sp_bar = dir_bar.z * st;
                double st_bar = dir_bar.x * cp + dir_bar.z * sp, ct_bar = dir_bar.y;
                double phi_bar = cp_bar * (-sp) + sp_bar * cp;
                double theta_bar = ct_bar * (-st) + st_bar * ct;
                adj_screen_tail(cam, screen_in, v2(phi_bar * double(2 * M_PI), theta_bar * double(M_PI)), g, want_screen_bar, screen_bar);
            }
        } break;
        default: {
            V3 pt = V3{(screen.x - 0.5f) * 2.f, (screen.y - 0.5f) * (-2.f) / ar, 1.0};
            V3 dir = mul(cam.intrinsic_mat_inv, pt);
            V3 dl = normalize(dir);
            V3 dw = xfm_vector(cam.cam_to_world, dl);
            V3 dw_bar = adj_normalize(dw, ray_bar.dir);
            V3 dl_bar = v3(0);
            adj_xfm_vector(cam.cam_to_world, dl, dw_bar, c2w_bar, dl_bar);
            V3 dir_bar = adj_normalize(dir, dl_bar);
            accum_outer3(g.intrinsic_mat_inv, dir_bar, pt, 3);
            V3 o_bar = v3(0);
            adj_xfm_point(cam.cam_to_world, v3(0), ray_bar.org, c2w_bar, o_bar);
            scatter_cam_to_world(cam, c2w_bar, g);
            if (want_screen) {
                V3 pt_bar = mul_t(cam.intrinsic_mat_inv, dir_bar);
                adj_screen_tail(cam, screen_in, v2(pt_bar.x * 2, pt_bar.y * (-2 / ar)), g, want_screen_bar, screen_bar);
            }
        } break;
    }
}

// Camera-space point -> screen position.
RDR_FN V2 camera_to_screen(const CameraD &cam, V3 pt) {
    double ar = aspect_of(cam);
    switch (cam.kind) {
        case kCamOrthographic: {
            V3 ip = mul(cam.intrinsic_mat, pt);
            return distort(cam.distortion, v2((ip.x + 1.f) * 0.5f, (-ip.y * ar + 1.f) * 0.5f));
        }
        case kCamFisheye: {
            V3 dir = normalize(pt);
            double phi = atan2(dir.y, dir.x);
            double theta = acos(dir.z);
            double r = theta * 2.f / double(M_PI);
            return distort(cam.distortion, v2(0.5f * (-r * cos(phi) + 1.f), 0.5f * (-r * sin(phi) + 1.f)));
        }
        case kCamPanorama: {
            V3 dir = normalize(pt);
            double phi = atan2(dir.z, dir.x);
            double theta = acos(dir.y);
            return distort(cam.distortion, v2(phi / double(2 * M_PI), theta / double(M_PI)));
        }
        default: {
            V3 ip = mul(cam.intrinsic_mat, pt);
            double ix = ip.x / ip.z, iy = ip.y / ip.z;
            return distort(cam.distortion, v2((ix + 1.f) * 0.5f, (-iy * ar + 1.f) * 0.5f));
        }
    }
}

RDR_FN void adj_camera_to_screen(const CameraD &cam, V3 pt, double sx_bar, double sy_bar, const GCamera &g, V3 &pt_bar) {
    double ar = aspect_of(cam);
    switch (cam.kind) {
        case kCamOrthographic: {
            V3 ip = mul(cam.intrinsic_mat, pt);
            V2 dxy = v2(0, 0);
            adj_distort(cam.distortion, v2((ip.x + 1.f) * 0.5f, (-ip.y * ar + 1.f) * 0.5f), v2(sx_bar, sy_bar), g.distortion, dxy);
            V3 ip_bar = V3{dxy.x * 0.5f, dxy.y * -0.5f * ar, 0.0};
            accum_outer3(g.intrinsic_mat, ip_bar, pt, 2);
            const M3 &im = cam.intrinsic_mat;
            pt_bar.x += ip_bar.x * im.m[0][0] + ip_bar.y * im.m[1][0];
            pt_bar.y += ip_bar.x * im.m[0][1] + ip_bar.y * im.m[1][1];
            pt_bar.z += ip_bar.x * im.m[0][2] + ip_bar.y * im.m[1][2];
        } break;
        case kCamFisheye: {
            V3 dir = normalize(pt);
            double phi = atan2(dir.y, dir.x);
            double theta = acos(dir.z);
            double r = theta * 2.f / double(M_PI);
            V2 dxy = v2(0, 0);
            adj_distort(cam.distortion, v2(0.5f * (-r * cos(phi) + 1.f), 0.5f * (-r * sin(phi) + 1.f)), v2(sx_bar, sy_bar),
                        g.distortion, dxy);
            double r_bar = -0.5f * (cos(phi) * dxy.x + sin(phi) * dxy.y);
            double phi_bar = 0.5f * r * sin(phi) * dxy.x - 0.5f * r * cos(phi) * dxy.y;
            double theta_bar = r_bar * (2.f / double(M_PI));
            double ct_bar = -theta_bar / sqrt(1.f - dir.z * dir.z);
            double at = dir.x * dir.x + dir.y * dir.y;
            V3 dir_bar = V3{-phi_bar * dir.y / at, phi_bar * dir.x / at, ct_bar};
            pt_bar += adj_normalize(pt, dir_bar);
        } break;
        case kCamPanorama: {
            V3 dir = normalize(pt);
            double phi = atan2(dir.z, dir.x);
            double theta = acos(dir.y);
            V2 dxy = v2(0, 0);
            adj_distort(cam.distortion, v2(phi / double(2 * M_PI), theta / double(M_PI)), v2(sx_bar, sy_bar), g.distortion, dxy);
            double phi_bar = dxy.x / double(2 * M_PI);
            double theta_bar = dxy.y / double(M_PI);
            double ct_bar = -theta_bar / sqrt(1.f - dir.y * dir.y);
            double at = dir.x * dir.x + dir.z * dir.z;
            V3 dir_bar = V3{-phi_bar * dir.z / at, ct_bar, phi_bar * dir.x / at};
            pt_bar += adj_normalize(pt, dir_bar);
        } break;
        default: {
            V3 ip = mul(cam.intrinsic_mat, pt);
            double ix = ip.x / ip.z, iy = ip.y / ip.z;
            V2 dxy = v2(0, 0);
            adj_distort(cam.distortion, v2((ix + 1.f) * 0.5f, (-iy * ar + 1.f) * 0.5f), v2(sx_bar, sy_bar), g.distortion, dxy);
            double ix_bar = dxy.x * 0.5f, iy_bar = dxy.y * -0.5f * ar;
            V3 ip_bar = V3{ix_bar / ip.z, iy_bar / ip.z, -(ix_bar * ix / ip.z + iy_bar * iy / ip.z)};
            accum_outer3(g.intrinsic_mat, ip_bar, pt, 3);
            pt_bar += mul_t(cam.intrinsic_mat, ip_bar);
        } break;
    }
}

// World segment (p0,p1) -> screen, clipped against z = clip_near.  False when fully behind.
RDR_FN bool project_segment(const CameraD &cam, V3 p0, V3 p1, V2 &s0, V2 &s1) {
    V3 l0 = xfm_point(cam.world_to_cam, p0), l1 = xfm_point(cam.world_to_cam, p1);
    double cn = cam.clip_near;
    if (l0.z < cn && l1.z < cn) return false;
    if (l0.z < cn) {
        V3 d = l0 - l1;
        double t = -(l1.z - cn) / d.z;
        l0 = l1 + t * d;
    } else if (l1.z < cn) {
        V3 d = l1 - l0;
        double t = -(l0.z - cn) / d.z;
        l1 = l0 + t * d;
    }
    s0 = camera_to_screen(cam, l0);
    s1 = camera_to_screen(cam, l1);
    return true;
}

RDR_FN M4 mul44(const M4 &a, const M4 &b) {
    M4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
        r.m[i][j] = s;
    }
    return r;
}

RDR_FN void adj_project_segment(const CameraD &cam, V3 p0, V3 p1, V2 s0_bar, V2 s1_bar,
                                const GCamera &g, V3 &p0_bar, V3 &p1_bar) {
    V3 l0 = xfm_point(cam.world_to_cam, p0), l1 = xfm_point(cam.world_to_cam, p1);
    double cn = cam.clip_near;
    if (l0.z < cn && l1.z < cn) return;
    V3 c0 = l0, c1 = l1;
    if (l0.z < cn) {
        V3 d = l0 - l1;
        c0 = l1 + (-(l1.z - cn) / d.z) * d;
    } else if (l1.z < cn) {
        V3 d = l1 - l0;
        c1 = l0 + (-(l0.z - cn) / d.z) * d;
    }
    V3 c0_bar = v3(0), c1_bar = v3(0);
    adj_camera_to_screen(cam, c0, s0_bar.x, s0_bar.y, g, c0_bar);
    adj_camera_to_screen(cam, c1, s1_bar.x, s1_bar.y, g, c1_bar);
    V3 l0_bar = v3(0), l1_bar = v3(0);
    if (l0.z < cn) {
        V3 d = l0 - l1;
        double t = -(l1.z + cn) / d.z;       // [quirk] '+' in the reference's adjoint (src/camera.h:776)
        l1_bar += c0_bar;
        double t_bar = dot(d, c0_bar);
        V3 d_bar = t * c0_bar;
        l1_bar.z += (-t_bar / d.z);
        d_bar.z -= t_bar * t / d.z;
        l0_bar += d_bar; l1_bar -= d_bar;
        l1_bar += c1_bar;
    } else if (l1.z < cn) {
        V3 d = l1 - l0;
        double t = -(l0.z + cn) / d.z;
        l0_bar += c1_bar;
        double t_bar = dot(d, c1_bar);
        V3 d_bar = t * c1_bar;
        l0_bar.z += (-t_bar / d.z);
        d_bar.z -= t_bar * t / d.z;
        l1_bar += d_bar; l0_bar -= d_bar;
        l0_bar += c0_bar;
    } else {
        l0_bar += c0_bar; l1_bar += c1_bar;
    }
    M4 w2c_bar = m4_zero();
    adj_xfm_point(cam.world_to_cam, p0, l0_bar, w2c_bar, p0_bar);
    adj_xfm_point(cam.world_to_cam, p1, l1_bar, w2c_bar, p1_bar);
    // d(cam_to_world) = -W^T d(world_to_cam) W^T
    M4 wt;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) wt.m[i][j] = cam.world_to_cam.m[j][i];
    M4 tmp = mul44(mul44(wt, w2c_bar), wt);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) tmp.m[i][j] = -tmp.m[i][j];
    scatter_cam_to_world(cam, tmp, g);
}

// Screen point -> camera space (z = 1 plane for the linear models, unit sphere for fisheye / panorama).
RDR_FN V3 screen_to_camera(const CameraD &cam, V2 screen_in) {
    V2 screen = inverse_distort(cam.distortion, screen_in);
    double ar = aspect_of(cam);
    switch (cam.kind) {
        case kCamOrthographic: {
            V3 pt = V3{(screen.x - 0.5f) * 2.f, (screen.y - 0.5f) * -2.f / ar, 1.0};
            V3 d = mul(cam.intrinsic_mat_inv, pt);
            return V3{d.x, d.y, 1.0};
        }
        case kCamFisheye: {
            double x = 2.f * (screen.x - 0.5f), y = 2.f * (screen.y - 0.5f);
            double r = sqrt(x * x + y * y);
            double phi = atan2(y, x);
            double theta = r * double(M_PI) / 2.f;
            double sp = sin(phi), cp = cos(phi), st = sin(theta), ct = cos(theta);
            return V3{-cp * st, -sp * st, ct};
        }
        case kCamPanorama: return panorama_dir(screen);
        default: {
            V3 pt = V3{(screen.x - 0.5f) * 2.f, (screen.y - 0.5f) * -2.f / ar, 1.0};
            V3 d = mul(cam.intrinsic_mat_inv, pt);
            return V3{d.x / d.z, d.y / d.z, 1.0};
        }
    }
}
RDR_FN void adj_screen_to_camera(const CameraD &cam, V2 screen_in, V3 o_bar, V2 &screen_bar) {
    V2 screen = inverse_distort(cam.distortion, screen_in);
    double ar = aspect_of(cam);
    V2 distorted_bar;
    switch (cam.kind) {
        case kCamOrthographic: {
            V3 pt_bar = mul_t(cam.intrinsic_mat_inv, V3{o_bar.x, o_bar.y, 0.0});
            distorted_bar = v2(pt_bar.x * 2, pt_bar.y * (-2) / ar);
        } break;
        case kCamFisheye: {
            double x = 2.f * (screen.x - 0.5f), y = 2.f * (screen.y - 0.5f);
            double r = sqrt(x * x + y * y);
            double phi = atan2(y, x);
            double theta = r * double(M_PI) / 2.f;
            double sp = sin(phi), cp = cos(phi), st = sin(theta), ct = cos(theta);
            double cp_bar = -o_bar.x * st, sp_bar = -o_bar.y * st;
            double st_bar = -(o_bar.x * cp + o_bar.y * sp), ct_bar = o_bar.z;
            double phi_bar = sp_bar * cp - cp_bar * sp;
            double theta_bar = st_bar * ct - ct_bar * st;
            double r_bar = theta_bar * (double(M_PI) / 2.f);
            double x_bar = phi_bar * (-y / (x * x + y * y));
            double y_bar = phi_bar * (x / (x * x + y * y));
            x_bar += (r_bar * (x / r));
            y_bar += (r_bar * (y / r));
            distorted_bar = v2(x_bar * 2, y_bar * 2);
        } break;
        case kCamPanorama: {
            double theta = double(M_PI) * screen.y, phi = double(2 * M_PI) * screen.x;
            double sp = sin(phi), cp = cos(phi), st = sin(theta), ct = cos(theta);
            double cp_bar = o_bar.x * st;
            double sp_bar = o_bar.z * sp;                 // [quirk] sin_phi where sin_theta is meant (src/camera.h:1018)
            double st_bar = o_bar.x * cp + o_bar.z * sp, ct_bar = o_bar.y;
            double phi_bar = sp_bar * cp - cp_bar * sp;
            double theta_bar = st_bar * ct - ct_bar * st;
            double x_bar = phi_bar * double(2 * M_PI), y_bar = theta_bar * double(M_PI);
            distorted_bar = v2(x_bar * 2, y_bar * 2);     // [quirk] extra factor 2 (src/camera.h:1034)
        } break;
        default: {
            V3 pt = V3{(screen.x - 0.5f) * 2.f, (screen.y - 0.5f) * -2.f / ar, 1.0};
            V3 d = mul(cam.intrinsic_mat_inv, pt);
            V3 dn = V3{d.x / d.z, d.y / d.z, 1.0};
            V3 d_bar = V3{o_bar.x / d.z, o_bar.y / d.z, -(o_bar.x * dn.x / d.z + o_bar.y * dn.y / d.z)};
            V3 pt_bar = mul_t(cam.intrinsic_mat_inv, d_bar);
            distorted_bar = v2(pt_bar.x * 2, pt_bar.y * (-2) / ar);
        } break;
    }
    adj_inverse_distort(cam.distortion, screen_in, distorted_bar, nullptr, screen_bar);
}

RDR_FN bool in_screen(const CameraD &cam, V2 pt) {
    int xi = int(pt.x * cam.width), yi = int(pt.y * cam.height);
    if (xi < cam.vp_x0 || xi >= cam.vp_x1 || yi < cam.vp_y0 || yi >= cam.vp_y1) return false;
    if (cam.kind != kCamFisheye) return pt.x >= 0.f && pt.x < 1.f && pt.y >= 0.f && pt.y < 1.f;
    double dist_sq = (pt.x - 0.5f) * (pt.x - 0.5f) + (pt.y - 0.5f) * (pt.y - 0.5f);
    return dist_sq < 0.25f;
}

// Whether the primary edge estimator works in screen space (Eq. 8 of the paper) or on the camera-space film.
RDR_FN bool linear_projection(const CameraD &cam) {
    return (cam.kind == kCamPerspective || cam.kind == kCamOrthographic) && !cam.distortion.defined;
}

} // namespace rdr
