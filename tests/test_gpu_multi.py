"""N > 1 from ONE process on a ONE-GPU box: lh_multi_* with the same device listed twice (two replicas, two host
threads, two streams, the real tile queue and the real device-to-device gather).  The sharded frames must equal
the unsharded ones bit for bit; the scene is built once."""
import numpy as np
import pytest

import lucille_amd as la
from lucille_amd import render
from oracle import pyoracle as po
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu


def load_multi(name, devices):
    g = load_golden(name)
    m = la.HipMulti(devices)
    a = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        for t in (m, a):
            t.add_mesh(g["pos%d" % k], g["idx%d" % k])
            if ("nrm%d" % k) in g.files:
                t.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
    m.commit(); a.commit()
    oc = po.Camera.from_ref(g["camera"])
    return g, m, a, oc


@pytest.mark.parametrize("name,ps,ns,tile", [("ao_c1", 1, 16, 64), ("ao_ps", 2, 9, 40)])
def test_sharded_ao_frame_equals_unsharded(name, ps, ns, tile):
    import torch
    g, m, a, oc = load_multi(name, [0, 0])
    assert m.n == 2
    cam = la.Camera.make(oc.width, oc.height, oc.flength, list(oc.cam2world), oc.rh)
    rgb, st, secs = m.render_ao_frame(cam, ps, ns, seed=9, tile=tile)
    ref, st1 = render.render_ao_frame(a, cam, ps, ns, tile=cam.width, seed=9)
    torch.cuda.synchronize()
    assert np.array_equal(rgb, ref.cpu().numpy())
    assert st == st1 and st["primary_hits"] > 0
    assert len(secs) == 2 and all(s > 0 for s in secs)          # both replicas drained tiles
    # one host build: the replica shares replica 0's scene
    i0, i1 = m.accel(0).info(), m.accel(1).info()
    assert i0["build_seconds"] == i1["build_seconds"] and i0["ntriangles"] == i1["ntriangles"] == a.info()["ntriangles"]
    # three replicas, ragged tiles
    m3 = la.HipMulti([0, 0, 0])
    for k in range(int(g["ngeoms"])):
        m3.add_mesh(g["pos%d" % k], g["idx%d" % k])
        if ("nrm%d" % k) in g.files:
            m3.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
    m3.commit()
    rgb3, st3, _ = m3.render_ao_frame(cam, ps, ns, seed=9, tile=37)
    assert np.array_equal(rgb3, rgb) and st3 == st
    m.close(); m3.close(); a.close()


def test_sharded_pt_frame_and_ray_dump_equal_unsharded():
    import torch
    g, m, a, oc = load_multi("ao_ps", [0, 0])
    cam = la.Camera.make(oc.width, oc.height, oc.flength, list(oc.cam2world), oc.rh)
    mat = la.Material.make(kd=(0.6, 0.5, 0.4), ks=(0.1, 0.1, 0.1), kt=(0.2, 0.2, 0.2), ior=1.3)
    m.set_material(la.ALL_MESHES, mat); a.set_material(la.ALL_MESHES, mat)
    m.set_environment((1.0, 0.9, 0.8), None); a.set_environment((1.0, 0.9, 0.8), None)
    rgb, st, _ = m.render_pt_frame(cam, 24, spp_chunk=8, max_vertices=6, seed=4, tile=32)
    ref = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    rays = 0
    for s0 in range(0, 24, 8):
        _, s1 = a.render_pt_tile2(cam, 0, 0, cam.width, cam.height, s0, 8, 24, max_vertices=6, seed=4, out=ref)
        rays += s1["rays"]
    torch.cuda.synchronize()
    # per-pixel sums of the same samples in the same order: equal up to float summation of 3 passes per tile either way
    assert np.allclose(rgb, ref.cpu().numpy(), rtol=0, atol=1e-6)
    assert st["rays"] == rays and st["paths"] == cam.width * cam.height * 24
    # ray dump: two contiguous slices
    P, idx, org, dr = po.soup(20000, 50001, 0.01, 77)
    m2 = la.HipMulti([0, 0]); m2.add_mesh(P, idx); m2.commit()
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    got = m2.intersect_host(org, dr)
    for x, y in zip(got, exp):
        assert np.array_equal(x, y)
    assert np.array_equal(m2.intersect_host(org, dr, mode=la.MODE_ANY).astype(bool), exp[0] != po.MISS)
    m.close(); m2.close(); a.close()


def test_replicas_of_a_device_built_scene_build_both_trees_themselves():
    """lh_multi_commit(LH_BUILD_ON_DEVICE): replica 0 builds the traversal tree AND lucille's own tree on its device, the other
    replicas do the same on theirs (there is no host copy to upload): frames, ray dumps with exact-t ties and the trees agree"""
    import torch
    from lucille_amd import scenes
    from tests.test_gpu_refbuild import assert_same_tree
    g = load_golden("ao_c1")
    m = la.HipMulti([0, 0]); a = la.HipAccel(0)
    meshes = [scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 3) for k in range(int(g["ngeoms"]))]
    for P, I in meshes:
        m.add_mesh(P, I); a.add_mesh(P, I)
    m.commit(-2); a.commit()
    oc = po.Camera.from_ref(g["camera"])
    cam = la.Camera.make(oc.width, oc.height, oc.flength, list(oc.cam2world), oc.rh)
    rgb, st, _ = m.render_ao_frame(cam, 1, 16, seed=5, tile=64)
    ref, st1 = render.render_ao_frame(a, cam, 1, 16, tile=cam.width, seed=5)
    torch.cuda.synchronize()
    assert np.array_equal(rgb, ref.cpu().numpy()) and st == st1
    th = a.ref_tree()
    for k in range(2):
        assert_same_tree(th, m.accel(k).ref_tree(), "replica %d" % k)
    m.close(); a.close()
