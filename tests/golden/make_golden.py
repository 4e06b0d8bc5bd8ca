#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ (run in the BUILD container only).

Needs /root/reference (the unmodified pyredner package, to parse tests/scenes/bunny_box.xml) and
the oracle build (oracle/_ref, `make -C oracle`).  Two kinds of files are written:

  bunny_box_scene.npz        raw mesh/material/camera arrays of the reference's bunny_box scene,
                             exactly as pyredner.load_mitsuba hands them to redner.Scene
                             (tests/test_bunny_box.py) -- INPUT data for tests and bench.py
  <case>.npz                 oracle outputs (forward image, every gradient tensor) for the parity
                             cases listed in CASES, rendered through tests/scenes.py with the
                             oracle as backend

The GPU box has neither /root/reference nor network; tests read only these files.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]

import oracle_util  # noqa: E402

REF = '/root/reference'


def export_bunny_box():
    ref = oracle_util.load_oracle()
    sys.modules['redner'] = ref
    sys.path[:0] = [os.path.join(ROOT, 'oracle', 'pystubs'), REF]
    import pyredner
    pyredner.set_use_gpu(False)
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, 'tests'))
    scene = pyredner.load_mitsuba('scenes/bunny_box.xml')
    os.chdir(cwd)
    out = {'num_shapes': len(scene.shapes), 'num_materials': len(scene.materials)}
    cam = scene.camera
    if cam.cam_to_world is not None:
        out['cam_to_world'] = cam.cam_to_world.numpy()
    else:
        out['cam_position'], out['cam_look_at'], out['cam_up'] = cam.position.numpy(), cam.look_at.numpy(), cam.up.numpy()
    out['intrinsic_mat'] = cam.intrinsic_mat.numpy()
    out['clip_near'] = cam.clip_near
    bunny = max(range(len(scene.shapes)), key=lambda i: scene.shapes[i].indices.shape[0])
    out['bunny_shape_id'] = bunny
    for i, sh in enumerate(scene.shapes):
        out['shape%d_vertices' % i] = sh.vertices.numpy()
        out['shape%d_indices' % i] = sh.indices.numpy().astype(np.int32)
        if sh.uvs is not None:
            out['shape%d_uvs' % i] = sh.uvs.numpy()
        if sh.normals is not None:
            out['shape%d_normals' % i] = sh.normals.numpy()
        assert sh.uv_indices is None and sh.normal_indices is None and sh.colors is None
        out['shape%d_material_id' % i] = sh.material_id
    for i, m in enumerate(scene.materials):
        assert m.diffuse_reflectance.texels.dim() == 1 and m.specular_reflectance.texels.dim() == 1
        assert m.roughness.texels.dim() == 1 and m.normal_map is None and m.generic_texture is None
        out['mat%d_diffuse' % i] = m.diffuse_reflectance.texels.numpy()
        out['mat%d_specular' % i] = m.specular_reflectance.texels.numpy()
        out['mat%d_roughness' % i] = m.roughness.texels.numpy()
        out['mat%d_compute_specular_lighting' % i] = m.compute_specular_lighting
        out['mat%d_two_sided' % i] = m.two_sided
    assert len(scene.area_lights) == 1 and scene.envmap is None
    l = scene.area_lights[0]
    out['light0_shape_id'], out['light0_intensity'], out['light0_two_sided'] = l.shape_id, l.intensity.numpy(), l.two_sided
    np.savez_compressed(os.path.join(HERE, 'bunny_box_scene.npz'), **out)
    print('bunny_box_scene.npz: %d shapes, %d triangles' % (len(scene.shapes), sum(s.indices.shape[0] for s in scene.shapes)))


# name -> (scene builder, resolution, spp, max_bounces)
CASES = {
    'single_triangle_64x64x4': ('single_triangle', 64, 4, 1),
    'two_triangles_64x64x16': ('two_triangles', 64, 16, 1),
    'bunny_box_32x32x4': ('bunny_box', 32, 4, 4),
}


def render_case(backend, builder, res, spp, mb, device=torch.device('cpu'), grad_mode='sum'):
    """Forward + backward of one case; returns {'image': ..., 'grad_<i>_<name>': ...}."""
    import scenes
    from redner_amd.render_pytorch import RenderFunction
    sc = getattr(scenes, builder)(device, resolution=(res, res))
    for l in sc.area_lights:
        l.intensity.requires_grad_(True)
    for m in sc.materials:
        m.diffuse_reflectance.mipmap[0].requires_grad_(True)
    if sc.camera.position is not None:
        sc.camera.position.requires_grad_(True)
    args = RenderFunction.serialize_scene(sc, spp, mb, sampler_type=backend.SamplerType.sobol, device=device,
                                          backend=backend)
    img = RenderFunction.apply(1, *args)
    out = {'image': img.detach().cpu().numpy()}
    # upstream gradient: a fixed smooth pattern so every pixel/channel has a distinct weight
    h, w, c = img.shape
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    up = torch.stack([1.0 + 0.5 * torch.sin(0.37 * xx + 0.11 * yy), 1.0 + 0.5 * torch.cos(0.23 * yy),
                      1.0 - 0.3 * torch.sin(0.19 * (xx + yy))], dim=2).to(img.device)
    (img * up).sum().backward()
    for i, sh in enumerate(sc.shapes):
        if sh.vertices.grad is not None:
            out['grad_shape%d_vertices' % i] = sh.vertices.grad.cpu().numpy()
    for i, l in enumerate(sc.area_lights):
        out['grad_light%d_intensity' % i] = l.intensity.grad.cpu().numpy()
    for i, m in enumerate(sc.materials):
        out['grad_mat%d_diffuse' % i] = m.diffuse_reflectance.mipmap[0].grad.cpu().numpy()
    if sc.camera.position is not None:
        out['grad_cam_position'] = sc.camera.position.grad.cpu().numpy()
    return out


def main():
    if not os.path.exists(os.path.join(HERE, 'bunny_box_scene.npz')) or '--scene' in sys.argv:
        export_bunny_box()
    ref = oracle_util.load_oracle()
    for name, (builder, res, spp, mb) in CASES.items():
        out = render_case(ref, builder, res, spp, mb)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
        print(name, {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
