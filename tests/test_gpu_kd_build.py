"""The device-side k-d builder (rt_build_kdtree_device) against the same pins as the host builder: the reference's
own arrays for the four bunny resolutions (fixtures captured from the reference's JavaScript), the oracle builder on
random meshes, and rendered frames that must not depend on where the tree was built."""
import numpy as np
import pytest

from helpers import make_flat, manifest, model, sha
from oracle import oracle
from ray_tracer_amd import abi, scene

pytestmark = pytest.mark.gpu

MESHES = [("bun_zipper_res4.ply", "k8_res4_640x480"), ("bun_zipper_res3.ply", "k8_res3_640x480"),
          ("bun_zipper_res2.ply", "k8_res2_640x480"), ("bun_zipper.ply", "k8_full_640x480")]


@pytest.mark.parametrize("ply,gold", MESHES)
def test_device_builder_bit_exact_against_reference(ply, gold):
    m = manifest(gold)
    sc = scene.default_scene(model(ply))
    v, f, fm = sc.mesh_arrays()
    rt = abi.RayTracerHIP(abi.default_config(64, 48))
    rt.setMesh(v, f, fm)
    rt.buildKdTree(on_device=True)
    pairs, leaves, box = rt.exportKdTree()
    kd = m["kd"]
    assert sha(box) == kd["sha256"]["root_box_f64"]
    types = pairs[:, 1] >> 30
    assert (int((types == 1).sum()), int((types == 3).sum()), int((types == 2).sum())) == (kd["n_internal"], kd["n_leaf"], kd["n_empty"])
    assert sha(leaves) == kd["sha256"]["leaves_i32"]
    assert sha(pairs) == kd["sha256"]["nodes_sparse_u32"]
    # a second build of the same mesh queues all levels at once over the sizes the first one left behind
    rt.buildKdTree(on_device=True)
    pairs2, leaves2, box2 = rt.exportKdTree()
    assert np.array_equal(pairs2, pairs) and np.array_equal(leaves2, leaves) and np.array_equal(box2, box)
    rt.close()


def test_device_builder_sums_in_two_steps(monkeypatch):
    """The exclusive sums inside the builder's kernels go in two steps (groups of 256 workgroups) for launches of more
    than 2048 workgroups, which none of the test meshes reaches. RT_KD_CHAIN_FLAT lowers that size: with 0 every launch
    goes the two-step way, and the full bunny (levels of up to 1065 workgroups, five groups) must still give the
    reference's arrays, on the first build (level by level) and on the second (all levels queued at once)."""
    monkeypatch.setenv("RT_KD_CHAIN_FLAT", "0")
    test_device_builder_bit_exact_against_reference("bun_zipper.ply", "k8_full_640x480")
    test_device_builder_random_meshes_match_oracle()
    test_device_builder_large_mesh_equals_host_builder()


def test_device_builder_large_mesh_equals_host_builder():
    """A height field of 500 000 faces: levels of more than 2048 workgroups (the sums inside the kernels then go in two
    steps without any test hook) and lists an order of magnitude longer than the bunny's. The oracle would need minutes
    for it; the check is the threaded host builder, which the CPU tests pin to the reference and to the oracle."""
    n = 501
    gx, gy = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64), indexing="ij")
    gz = 12.0 * np.sin(gx * 0.045) * np.cos(gy * 0.031) + 3.0 * np.sin(gx * 0.21 + gy * 0.17)
    v = np.stack([gx * 0.4 - 100.0, gz, gy * 0.4 - 100.0], axis=-1).reshape(-1, 3)
    idx = (np.arange(n - 1)[:, None] * n + np.arange(n - 1)[None, :]).reshape(-1)
    f = np.concatenate([np.stack([idx, idx + 1, idx + n], axis=1), np.stack([idx + 1, idx + n + 1, idx + n], axis=1)]).astype(np.uint32)
    assert len(f) == 500000
    fm = np.zeros((len(f), 2), np.uint32)
    rt = abi.RayTracerHIP(abi.default_config(64, 48))
    rt.setMesh(v, f, fm)
    rt.buildKdTree(on_device=False)
    hp, hl, hb = rt.exportKdTree()
    for _ in range(2):   # level by level, then all levels queued at once
        rt.setMesh(v, f, fm)
        rt.buildKdTree(on_device=True)
        pairs, leaves, box = rt.exportKdTree()
        assert np.array_equal(hb, box)
        assert np.array_equal(hl, leaves)
        assert np.array_equal(hp, pairs)
    rt.close()


def test_device_builder_lists_grow_for_a_flat_mesh():
    """120 faces in one plane: the root box has no extent on one axis, a third of the levels separate nothing, and the
    tree grows to millions of nodes. The builder's lists, sized from the face count, run out several times over (faces
    per level, nodes per level, leaves); each time the list that ran out is made longer and the build starts again. The
    check is the host builder (the oracle would need minutes)."""
    rng = np.random.default_rng(21)
    n_faces = 120
    v = rng.uniform(-50, 50, size=(n_faces * 2, 3))
    f = rng.integers(0, len(v), size=(n_faces, 3)).astype(np.uint32)
    v[f[:, 1]] = v[f[:, 0]] + rng.normal(0, 2.0, size=(n_faces, 3))
    v[f[:, 2]] = v[f[:, 0]] + rng.normal(0, 2.0, size=(n_faces, 3))
    v[:, 1] = 1.5
    fm = np.zeros((n_faces, 2), np.uint32)
    rt = abi.RayTracerHIP(abi.default_config(64, 48))
    rt.setMesh(v, f, fm)
    rt.buildKdTree(on_device=False)
    hp, hl, hb = rt.exportKdTree()
    assert len(hp) > 1000000
    for _ in range(2):
        rt.setMesh(v, f, fm)
        rt.buildKdTree(on_device=True)
        pairs, leaves, box = rt.exportKdTree()
        assert np.array_equal(hb, box) and np.array_equal(hl, leaves) and np.array_equal(hp, pairs)
    rt.close()


def test_device_builder_random_meshes_match_oracle():
    rng = np.random.default_rng(11)
    rt = abi.RayTracerHIP(abi.default_config(64, 48))
    for n_faces in (5, 6, 40, 700, 3000):
        v = rng.uniform(-50, 50, size=(n_faces * 2, 3))
        f = rng.integers(0, len(v), size=(n_faces, 3)).astype(np.uint32)
        v[f[:, 1]] = v[f[:, 0]] + rng.normal(0, 2.0, size=(n_faces, 3))
        v[f[:, 2]] = v[f[:, 0]] + rng.normal(0, 2.0, size=(n_faces, 3))
        rt.setMesh(v, f, np.zeros((n_faces, 2), np.uint32))
        rt.buildKdTree(on_device=True)
        pairs, leaves, box = rt.exportKdTree()
        op, ol, ob, _ = oracle.kd_build(v, f)
        assert np.array_equal(ob, box), n_faces
        assert np.array_equal(ol, leaves), n_faces
        assert np.array_equal(op, pairs), n_faces
    # small inputs fall through to the host builder
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float64)
    f = np.array([[0, 1, 2], [0, 1, 3], [1, 2, 3]], np.uint32)
    rt.setMesh(v, f, np.zeros((3, 2), np.uint32))
    rt.buildKdTree(on_device=True)
    pairs, leaves, _ = rt.exportKdTree()
    assert pairs.tolist() == [[0, (3 << 30) | 0]] and leaves.tolist() == [[0, 1, 2, -1]]
    rt.close()


def test_device_builder_repeated_builds_with_predicted_sizes():
    """Builds after the first one of a face count run without waiting for the sizes of the levels: the same mesh
    again, the mesh moved, a mesh of the same face count whose tree has quite different level sizes (the prediction
    fails and the build is repeated level by level), and back. Every tree must equal the oracle's."""
    rng = np.random.default_rng(5)
    n_faces = 3000

    def spread():
        v = rng.uniform(-50, 50, size=(n_faces * 2, 3))
        f = rng.integers(0, len(v), size=(n_faces, 3)).astype(np.uint32)
        v[f[:, 1]] = v[f[:, 0]] + rng.normal(0, 2.0, size=(n_faces, 3))
        v[f[:, 2]] = v[f[:, 0]] + rng.normal(0, 2.0, size=(n_faces, 3))
        return v, f

    va, fa = spread()
    vb, fb = spread()
    vb = vb * np.array([0.02, 0.02, 0.02])          # nearly everything in one small cluster ...
    vb[:40] = rng.uniform(-500, 500, size=(40, 3))   # ... inside a huge root box: much deeper, much smaller levels
    rt = abi.RayTracerHIP(abi.default_config(64, 48))
    for v, f in ((va, fa), (va, fa), (va + np.array([3.0, -2.0, 5.0]), fa), (vb, fb), (vb, fb), (va, fa)):
        rt.setMesh(v, f, np.zeros((n_faces, 2), np.uint32))
        rt.buildKdTree(on_device=True)
        pairs, leaves, box = rt.exportKdTree()
        op, ol, ob, _ = oracle.kd_build(v, f)
        assert np.array_equal(ob, box)
        assert np.array_equal(ol, leaves)
        assert np.array_equal(op, pairs)
    rt.close()


def test_frames_do_not_depend_on_where_the_tree_was_built():
    """Same scene, tree built by the host builder and by the device builder: identical float colours, bytes, hit ids
    and ray counts (the node layouts differ, the trees do not). Then the mesh moves and the device tree is rebuilt in
    the same context, as a per-frame rebuild does."""
    flat = make_flat("bun_zipper_res3.ply", 320, 240, depth=3, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    outs = []
    for dev in (False, True):
        rt = abi.RayTracerHIP(flat.cfg)
        rt.setScene(flat, kd_on_device=dev)
        rt.trace()
        outs.append((rt.readFloat(), rt.readRGBA8(), rt.readHitIds(), rt.stats().rays_total, rt.stats().shadow_rays, rt))
    assert np.array_equal(outs[0][0], outs[1][0], equal_nan=True)
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert outs[0][3:5] == outs[1][3:5]
    moved = flat.verts + np.array([3.0, -2.0, 5.0])
    for i, dev in enumerate((False, True)):
        rt = outs[i][5]
        rt.setMesh(moved, flat.faces, flat.face_mat)
        rt.buildKdTree(on_device=dev)
        rt.trace()
    a, b = outs[0][5], outs[1][5]
    assert np.array_equal(a.readRGBA8(), b.readRGBA8()) and np.array_equal(a.readHitIds(), b.readHitIds())
    assert not np.array_equal(a.readRGBA8(), outs[0][1])      # the mesh did move
    pa, la, ba = a.exportKdTree()
    pb, lb, bb = b.exportKdTree()
    assert np.array_equal(pa, pb) and np.array_equal(la, lb) and np.array_equal(ba, bb)


def test_animated_frames_with_rebuild_match_the_oracle():
    """benchmark configuration 5 in small: light 0 on its orbit, the mesh dragged with Mesh.translate (float32 matrix round trip
    of every vertex, as the reference's drag does it), the tree rebuilt on the device for every frame, three frames in
    flight on alternating contexts. Every frame must equal the oracle's render of that frame's scene (the oracle builds
    its own tree from the moved vertices)."""
    import torch
    from helpers import dragged_frames
    from ray_tracer_amd import multi_gpu
    frames = [0, 30, 60, 90, 120, 150]
    flats = dragged_frames("bun_zipper_res4.ply", frames, 160, 120, 3, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    flat = flats[0]
    assert not np.array_equal(flats[30].verts, flats[0].verts)
    rts = []
    for _ in range(3):
        rt = abi.RayTracerHIP(flat.cfg)
        rt.setScene(flat, kd_on_device=True)
        rts.append(rt)
    job = multi_gpu.PipelinedFrameJob(rts, multi_gpu.Comm(0, 1, 0, backend=None, device="cuda:0"))
    for k in frames:
        f = flats[k]
        rt = job.next_context()
        rt.setLights(f.lights, f.n_lights_loop)
        rt.setMesh(f.verts, f.faces, f.face_mat)
        rt.buildKdTree(on_device=True)
        job.step()
    job.flush()
    torch.cuda.synchronize()
    for slot, k in ((0, frames[3]), (1, frames[4]), (2, frames[5])):   # the last frame of each context
        want_f, want_8, want_id, cn = oracle.OracleRenderer(flats[k]).trace(threads=16)
        rt = rts[slot]
        assert np.array_equal(rt.readHitIds(), want_id), k
        assert np.array_equal(rt.readRGBA8(), want_8), k
        assert np.array_equal(rt.readFloat(), want_f, equal_nan=True), k
        assert rt.stats().rays_total == cn.rays
    job.close()


def test_one_animated_frame_at_full_size_matches_the_oracle():
    """The animated benchmark at its real size: full bunny, 3840x2160, depth 4, frame 37 of the run (light 0 on its
    orbit, the mesh moved along its circle with Mesh.translate), the tree rebuilt on the device - after a first build at
    rest, so that this is a rebuild over predicted sizes. The whole frame must equal the oracle's render of the moved scene."""
    from helpers import dragged_frames
    flats = dragged_frames("bun_zipper.ply", [0, 37], 3840, 2160, 4)
    flat, f = flats[0], flats[37]
    rt = abi.RayTracerHIP(flat.cfg)
    rt.setScene(flat, kd_on_device=True)
    rt.trace()
    rt.setLights(f.lights, f.n_lights_loop)
    rt.setMesh(f.verts, f.faces, f.face_mat)
    rt.buildKdTree(on_device=True)
    rt.trace()
    want_f, want_8, want_id, cn = oracle.OracleRenderer(f).trace(threads=16)
    assert np.array_equal(rt.readHitIds(), want_id)
    assert np.array_equal(rt.readRGBA8(), want_8)
    assert np.array_equal(rt.readFloat(), want_f, equal_nan=True)
    st = rt.stats()
    assert st.rays_total == cn.rays and st.shadow_rays == cn.shadow_rays
    rt.close()


def test_config5_with_the_project_scenes_supersampling_2_lattice_parity():
    """benchmark configuration 5 as project.js is configured: SUPERSAMPLING 2, i.e. a 7680x4320 render target behind a 3840x2160
    canvas, depth 4, an animated frame (mesh moved with Mesh.translate, tree rebuilt on the device). The oracle traces a
    lattice of every 12th pixel in x and y (230 400 pixels) and two 64x64 windows on the bunny; hit ids, bytes and float
    colours must be equal there."""
    from helpers import dragged_frames
    flats = dragged_frames("bun_zipper.ply", [0, 23], 3840, 2160, 4, supersampling=2)
    f = flats[23]
    rt = abi.RayTracerHIP(f.cfg)
    rt.setScene(f, kd_on_device=True)
    rt.trace()
    a8, af, aid = rt.readRGBA8(), rt.readFloat(), rt.readHitIds()
    assert a8.shape == (4320, 7680, 4) and rt.stats().pixels == 7680 * 4320
    o = oracle.OracleRenderer(f)
    wf, w8, wid, _ = o.trace(step=12, threads=16)
    lat = (slice(0, None, 12), slice(0, None, 12))
    assert np.array_equal(aid[lat], wid[lat]) and np.array_equal(a8[lat], w8[lat])
    assert ((af[lat] == wf[lat]) | (np.isnan(af[lat]) & np.isnan(wf[lat]))).all()
    n_prims = len(f.prims)
    ys, xs = np.nonzero(aid >= n_prims)
    assert len(ys) > 100000                                   # the mesh fills a good part of the frame
    cy, cx = int(np.median(ys)), int(np.median(xs))
    for rect in ((cx - 32, cy - 32, cx + 32, cy + 32), (int(xs.min()), cy - 32, int(xs.min()) + 64, cy + 32)):
        wf, w8, wid, _ = o.trace(rect=rect, threads=16)
        sl = (slice(rect[1], rect[3]), slice(rect[0], rect[2]))
        assert np.array_equal(aid[sl], wid[sl]) and np.array_equal(a8[sl], w8[sl]), rect
        assert ((af[sl] == wf[sl]) | (np.isnan(af[sl]) & np.isnan(wf[sl]))).all(), rect
    rt.close()


@pytest.mark.parametrize("on_device", [False, True])
def test_moved_vertices_under_the_old_tree_match_the_oracle(on_device):
    """rt_update_vertices: new vertices through the tree and root box as they stand (what the reference renders between
    two k-d rebuilds, js/8-kd_tree.js:2459-2533). The mesh is moved far enough that many faces leave the cells that list
    them: the walk's pruning margin (rt_kdtree_overhang) must follow, and the frames - 640x480, depth 4, reflective and
    transmissive mesh - must equal the oracle's for (moved vertices, old tree). A rebuild afterwards starts from the
    moved vertices."""
    import numpy as np
    from oracle import oracle
    flat = make_flat("bun_zipper_res4.ply", 640, 480, depth=4, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    rt = abi.RayTracerHIP(flat.cfg)
    rt.setScene(flat, kd_on_device=on_device)
    rt.trace()
    kd0 = oracle.kd_build(flat.verts, np.asarray(flat.faces, np.uint32))
    over0 = rt.kdOverhang()
    v0 = np.asarray(flat.verts, np.float64)
    rng = np.random.default_rng(5)
    for k, shift in enumerate(((1.5, 0.75, -1.0), (9.0, -4.0, 6.0), (-20.0, 3.0, 2.0))):
        # a translation plus a little noise per vertex: no two frames share a vertex array
        v = v0 + np.array(shift) + rng.uniform(-0.05, 0.05, v0.shape)
        rt.updateVertices(v)
        rt.trace()
        over = rt.kdOverhang()
        assert over.max() > over0.max()
        flat.verts = v
        o = oracle.OracleRenderer(flat, kd=kd0)
        want_f, want_8, want_id, cn = o.trace(threads=16)
        assert np.array_equal(rt.readHitIds(), want_id), k
        assert np.array_equal(rt.readRGBA8(), want_8), k
        got_f = rt.readFloat()
        assert ((got_f == want_f) | (np.isnan(got_f) & np.isnan(want_f))).all(), k
        assert rt.stats().rays_total == cn.rays
    # the next rebuild uses the moved vertices
    rt.buildKdTree(on_device=on_device)
    rt.trace()
    want = oracle.OracleRenderer(flat).trace(threads=16)
    assert np.array_equal(rt.readRGBA8(), want[1]) and np.array_equal(rt.readHitIds(), want[2])
    pairs, leaves, box = rt.exportKdTree()
    kd1 = oracle.kd_build(flat.verts, np.asarray(flat.faces, np.uint32))
    assert np.array_equal(pairs, kd1[0]) and np.array_equal(leaves, kd1[1]) and np.array_equal(box, kd1[2])
    rt.close()


def test_update_vertices_leaves_the_other_holders_of_a_shared_tree_alone():
    """Two contexts with the same scene share one device copy of the tree; moving the vertices of one must not move the
    other's."""
    import numpy as np
    flat = make_flat("bun_zipper_res4.ply", 320, 240, depth=2)
    a, b = abi.RayTracerHIP(flat.cfg), abi.RayTracerHIP(flat.cfg)
    a.setScene(flat)
    b.setScene(flat)
    a.trace()
    b.trace()
    before = b.readRGBA8()
    assert np.array_equal(a.readRGBA8(), before)
    a.updateVertices(np.asarray(flat.verts, np.float64) + np.array([12.0, 5.0, -3.0]))
    a.trace()
    b.trace()
    assert np.array_equal(b.readRGBA8(), before)
    assert not np.array_equal(a.readRGBA8(), before)
    a.close()
    b.close()


def test_update_vertices_refuses_another_vertex_count():
    flat = make_flat("bun_zipper_res4.ply", 64, 48)
    rt = abi.RayTracerHIP(flat.cfg)
    rt.setScene(flat)
    with pytest.raises(abi.RTError, match="another number of vertices"):
        rt.updateVertices(np.zeros((5, 3)))
    rt.close()


def test_a_stalled_device_build_falls_back_to_the_host_builder(monkeypatch, capfd):
    """The in-kernel sums of the device builder wait for workgroups with a lower index; the wait is bounded
    (RT_KD_CHAIN_POLL_LIMIT polls). With a limit of 0 the first count that is not there yet ends the build as
    stalled: rt_build_kdtree_device must say so, build the same tree on the host, and count the event - or, with
    RT_KD_NO_FALLBACK, report the stall as the error it is. Without the limit the same context builds on the device again."""
    m = manifest("k8_res2_640x480")
    flat = make_flat("bun_zipper_res2.ply", 320, 240, depth=2)
    ref = abi.RayTracerHIP(flat.cfg)
    ref.setScene(flat)
    ref.trace()
    want = ref.readRGBA8()
    rt = abi.RayTracerHIP(flat.cfg)
    monkeypatch.setenv("RT_KD_CHAIN_POLL_LIMIT", "0")
    rt.setScene(flat, kd_on_device=True)
    assert "building this tree on the host instead" in capfd.readouterr().err
    assert rt.stats().kd_device_fallbacks == 1
    pairs, leaves, box = rt.exportKdTree()
    assert sha(pairs) == m["kd"]["sha256"]["nodes_sparse_u32"] and sha(leaves) == m["kd"]["sha256"]["leaves_i32"]
    rt.trace()
    assert np.array_equal(rt.readRGBA8(), want)
    monkeypatch.setenv("RT_KD_NO_FALLBACK", "1")
    with pytest.raises(abi.RTError, match="never reported"):
        rt.buildKdTree(on_device=True)
    monkeypatch.delenv("RT_KD_NO_FALLBACK")
    monkeypatch.delenv("RT_KD_CHAIN_POLL_LIMIT")
    rt.buildKdTree(on_device=True)
    assert rt.stats().kd_device_fallbacks == 1
    pairs, leaves, box = rt.exportKdTree()
    assert sha(pairs) == m["kd"]["sha256"]["nodes_sparse_u32"] and sha(leaves) == m["kd"]["sha256"]["leaves_i32"]
    rt.trace()
    assert np.array_equal(rt.readRGBA8(), want)
    rt.close()
    ref.close()
