// camera.h -- pinhole camera: primary rays, projection of world segments to the screen, and their
// adjoints.
//
// Behavioural spec: sample_primary src/camera.h:121-141 (perspective branch), d_sample_primary_ray
// :199-277, camera_to_screen :508-521, project :561-591, d_camera_to_screen :595-640, d_project
// :731-830, screen_to_camera :832-851, d_screen_to_camera :899-934, in_screen :1049-1067; primary
// ray + finite-difference ray differential src/camera.cpp:8-43.
// Non-pinhole models (orthographic / fisheye / panorama) and lens distortion are the "next" row 3
// of SURVEY.md section 8f; Scene construction rejects them for now.
#pragma once
#include "surface.h"

namespace rdr {

RDR_FN double aspect_of(const CameraD &cam) { return double(cam.width) / double(cam.height); }

RDR_FN Ray make_ray(V3 org, V3 dir) { return Ray{org, dir, double(1e-3f), INFINITY}; }

RDR_FN Ray primary_ray(const CameraD &cam, V2 screen) {
    V3 org = xfm_point(cam.cam_to_world, v3(0));
    double ar = aspect_of(cam);
    V3 pt = V3{(screen.x - 0.5f) * 2.f, (screen.y - 0.5f) * (-2.f) / ar, 1.0};
    V3 dl = normalize(mul(cam.intrinsic_mat_inv, pt));
    V3 dw = normalize(xfm_vector(cam.cam_to_world, dl));
    return make_ray(org, dw);
}

// Screen position of sample `s` in pixel `pixel` of the viewport.
RDR_FN V2 pixel_to_screen(const CameraD &cam, int pixel, V2 s) {
    int vw = cam.vp_x1 - cam.vp_x0;
    int px = pixel % vw + cam.vp_x0, py = pixel / vw + cam.vp_y0;
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
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) accum(g.cam_to_world + 4 * r + c, c2w_bar.m[r][c]);
    }
}

// Adjoint of primary_ray(): pushes ray_bar into the camera gradient; optionally returns the
// screen-position adjoint.
RDR_FN void adj_primary_ray(const CameraD &cam, V2 screen, const DRay &ray_bar, const GCamera &g, V2 *screen_bar) {
    double ar = aspect_of(cam);
    V3 pt = V3{(screen.x - 0.5f) * 2.f, (screen.y - 0.5f) * (-2.f) / ar, 1.0};
    V3 dir = mul(cam.intrinsic_mat_inv, pt);
    V3 dl = normalize(dir);
    V3 dw = xfm_vector(cam.cam_to_world, dl);
    V3 dw_bar = adj_normalize(dw, ray_bar.dir);
    V3 dl_bar = v3(0);
    M4 c2w_bar = m4_zero();
    adj_xfm_vector(cam.cam_to_world, dl, dw_bar, c2w_bar, dl_bar);
    V3 dir_bar = adj_normalize(dir, dl_bar);
    if (g.intrinsic_mat_inv) {
        double db[3] = {dir_bar.x, dir_bar.y, dir_bar.z}, pv[3] = {pt.x, pt.y, pt.z};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) accum(g.intrinsic_mat_inv + 3 * r + c, db[r] * pv[c]);
    }
    V3 o_bar = v3(0);
    adj_xfm_point(cam.cam_to_world, v3(0), ray_bar.org, c2w_bar, o_bar);
    scatter_cam_to_world(cam, c2w_bar, g);
    if (screen_bar) {
        V3 pt_bar = mul_t(cam.intrinsic_mat_inv, dir_bar);
        screen_bar->x += pt_bar.x * 2;
        screen_bar->y += pt_bar.y * (-2 / ar);
    }
}

RDR_FN V2 camera_to_screen(const CameraD &cam, V3 pt) {
    double ar = aspect_of(cam);
    V3 ip = mul(cam.intrinsic_mat, pt);
    double ix = ip.x / ip.z, iy = ip.y / ip.z;
    return v2((ix + 1.f) * 0.5f, (-iy * ar + 1.f) * 0.5f);
}

RDR_FN void adj_camera_to_screen(const CameraD &cam, V3 pt, double sx_bar, double sy_bar, const GCamera &g, V3 &pt_bar) {
    double ar = aspect_of(cam);
    V3 ip = mul(cam.intrinsic_mat, pt);
    double ix = ip.x / ip.z, iy = ip.y / ip.z;
    double ix_bar = sx_bar * 0.5f, iy_bar = sy_bar * -0.5f * ar;
    V3 ip_bar = V3{ix_bar / ip.z, iy_bar / ip.z, -(ix_bar * ix / ip.z + iy_bar * iy / ip.z)};
    if (g.intrinsic_mat) {
        double ib[3] = {ip_bar.x, ip_bar.y, ip_bar.z}, pv[3] = {pt.x, pt.y, pt.z};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) accum(g.intrinsic_mat + 3 * r + c, ib[r] * pv[c]);
    }
    pt_bar += mul_t(cam.intrinsic_mat, ip_bar);
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

// Screen point -> camera-space point on the z = 1 plane.
RDR_FN V3 screen_to_camera(const CameraD &cam, V2 screen) {
    double ar = aspect_of(cam);
    V3 pt = V3{(screen.x - 0.5f) * 2.f, (screen.y - 0.5f) * -2.f / ar, 1.0};
    V3 d = mul(cam.intrinsic_mat_inv, pt);
    return V3{d.x / d.z, d.y / d.z, 1.0};
}
RDR_FN void adj_screen_to_camera(const CameraD &cam, V2 screen, V3 o_bar, V2 &screen_bar) {
    double ar = aspect_of(cam);
    V3 pt = V3{(screen.x - 0.5f) * 2.f, (screen.y - 0.5f) * -2.f / ar, 1.0};
    V3 d = mul(cam.intrinsic_mat_inv, pt);
    V3 dn = V3{d.x / d.z, d.y / d.z, 1.0};
    V3 d_bar = V3{o_bar.x / d.z, o_bar.y / d.z, -(o_bar.x * dn.x / d.z + o_bar.y * dn.y / d.z)};
    V3 pt_bar = mul_t(cam.intrinsic_mat_inv, d_bar);
    screen_bar.x += pt_bar.x * 2;
    screen_bar.y += pt_bar.y * (-2) / ar;
}

RDR_FN bool in_screen(const CameraD &cam, V2 pt) {
    int xi = int(pt.x * cam.width), yi = int(pt.y * cam.height);
    if (xi < cam.vp_x0 || xi >= cam.vp_x1 || yi < cam.vp_y0 || yi >= cam.vp_y1) return false;
    return pt.x >= 0.f && pt.x < 1.f && pt.y >= 0.f && pt.y < 1.f;
}

} // namespace rdr
