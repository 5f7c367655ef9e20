"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same scenes.

Bar: primary hit ids identical, pre-tone RGBA8 identical, float colour identical (the kernels and the
oracle follow one arithmetic contract: no FMA, IEEE divide/sqrt, the same deterministic pow), ray
counts identical; Lbar/sf to 1e-10 relative: the reference adds log(L)/N term by term in pixel order in
double precision, which carries ~N*2^-53 of rounding noise (3e-12 at 30k pixels, 1e-9 at 4K); the GPU
reduction sums in a different order."""
import numpy as np
import pytest

from helpers import device_canvas, make_flat, manifest, model, sha
from oracle import oracle
from ray_tracer_amd import abi, scene

pytestmark = pytest.mark.gpu
TONE_RTOL = 1e-10


def render_hip(flat):
    rt = abi.RayTracerHIP(flat.cfg)
    rt.setScene(flat)
    rt.trace()
    return rt


def check_against_oracle(flat, rect=None, threads=16):
    rt = render_hip(flat)
    got_f, got_8, got_id = rt.readFloat(), rt.readRGBA8(), rt.readHitIds()
    o = oracle.OracleRenderer(flat)
    want_f, want_8, want_id, cn = o.trace(rect=rect, threads=threads)
    if rect:
        x0, y0, x1, y1 = rect
        sl = (slice(y0, y1), slice(x0, x1))
    else:
        sl = (slice(None), slice(None))
    ids_ok = got_id[sl] == want_id[sl]
    assert ids_ok.all(), "hit ids differ at %d pixels" % (~ids_ok).sum()
    d8 = np.abs(got_8[sl].astype(int) - want_8[sl].astype(int))
    assert d8.max() == 0, "RGBA8 differs at %d pixels, max %d" % ((d8.max(axis=-1) > 0).sum(), d8.max())
    same_f = (got_f[sl] == want_f[sl]) | (np.isnan(got_f[sl]) & np.isnan(want_f[sl]))
    assert same_f.all(), "float colour differs at %d values" % (~same_f).sum()
    st = rt.stats()
    if not rect:
        assert st.rays_total == cn.rays and st.shadow_rays == cn.shadow_rays
        assert list(st.rays_per_level)[:5] == list(cn.rays_level)[:5]
    return rt, o, (got_f, got_8, got_id), cn


def test_config1_checkpoint6_scene_depth5():
    """benchmark configuration 1: checkpoint-6 scene (analytic primitives, one set light of two, INFINITY 100000.1)."""
    flat = make_flat(None, 640, 480, depth=5, time=None, second_light=False, infinity=100000.1)
    check_against_oracle(flat)


def test_default_scene_res4_depth1():
    flat = make_flat("bun_zipper_res4.ply", 640, 480, depth=1)
    check_against_oracle(flat)


def test_default_scene_res4_depth5_reflective_mesh():
    flat = make_flat("bun_zipper_res4.ply", 640, 480, depth=5, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    check_against_oracle(flat)


def test_config2_res4_1920x1080():
    """benchmark configuration 2: res4 bunny, primary rays + Phong, 1920x1080, light at t = 7.5 s."""
    flat = make_flat("bun_zipper_res4.ply", 1920, 1080, depth=1, time=7.5)
    check_against_oracle(flat)


def test_duplicate_faces_first_visited_wins():
    """Every face of the mesh twice, in a seeded random order, the copies with another material: a ray then finds
    two faces at exactly the same distance, often listed in different leaves (leaves keep four faces and drop the
    rest). The shader keeps the face it tests first (strict `hitInfo.t < tMin`, js/8-kd_tree.js:948-949): hit ids,
    colours and ray counts must still equal the oracle's, which walks the cells in the shader's order."""
    flat = make_flat("bun_zipper_res4.ply", 320, 240, depth=3, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    n = len(flat.faces)
    order = np.random.RandomState(7).permutation(2 * n)
    faces = np.concatenate([flat.faces, flat.faces])[order]
    mats = np.concatenate([flat.face_mat, np.tile(np.array([[0, 1]], flat.face_mat.dtype), (n, 1))])[order]
    flat.faces, flat.face_mat = np.ascontiguousarray(faces), np.ascontiguousarray(mats)
    _, _, (_, _, ids), _ = check_against_oracle(flat)
    mesh_ids = ids[ids >= len(flat.prims)] - len(flat.prims)
    assert len(mesh_ids) > 1000
    copies = (order[mesh_ids] >= n).mean()
    assert 0.2 < copies < 0.8, copies      # both the originals and the copies win somewhere


def test_brute_force_triangles_match():
    flat = make_flat("bun_zipper_res4.ply", 160, 120, depth=2, use_kd_tree=0)
    check_against_oracle(flat)


def test_ragged_frame_and_supersampling():
    # width/height not multiples of the 8x8 tile; 2x2 supersampling -> 202 x 150 target
    flat = make_flat("bun_zipper_res4.ply", 101, 75, depth=3, ss=2)
    rt, o, (gf, g8, gid), cn = check_against_oracle(flat)
    Lbar, sf = rt.tone()
    wl, ws = o.tone(g8)
    assert abs(Lbar - wl) <= TONE_RTOL * wl and abs(sf - ws) <= TONE_RTOL * ws
    assert np.array_equal(rt.present(), o.present(g8, wl, ws))
    flat.cfg.tr_method = 1
    rt2 = render_hip(flat)
    rt2.tone()
    assert np.array_equal(rt2.present(), oracle.OracleRenderer(flat).present(g8, wl, ws))


def test_odd_frame_rays_with_zero_direction_components():
    """An odd width and height put a pixel centre exactly on the camera axis: the rays of the middle column have
    direction x == 0, those of the middle row y == 0 (the corner rays are symmetric). Their plane distances on that axis
    are infinite or NaN (js/8-kd_tree.js:888-930 divides by the component); the walk must take the same branches as the
    shader, and such rays never hand stretches to idle lanes."""
    flat = make_flat("bun_zipper_res3.ply", 321, 241, depth=3, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    rt, o, (gf, g8, gid), cn = check_against_oracle(flat)
    assert rt.stats().rays_total == cn.rays


def test_moved_camera_pose_from_the_reference_controls():
    """The camera after a pan, a rotation and a wheel zoom with the reference's TrackballControls (fixture
    k8_res4_trackball_640x480: cameraPos and the four corner rays as the reference handed them to WebGL): the corner
    rays are no longer symmetric and the camera no longer sits at the origin. Depth 3 with the reflective mesh."""
    from helpers import manifest
    m = manifest("k8_res4_trackball_640x480")
    last = {}
    for u in m["uniforms"]:
        last[u["name"]] = u["value"]
    cam = {k: np.array(last[k], np.float32) for k in ("cameraPos", "ray00", "ray01", "ray10", "ray11")}
    assert np.abs(cam["cameraPos"]).max() > 10.0 and not np.allclose(cam["ray00"][:2], -cam["ray11"][:2])
    sc = scene.default_scene(model("bun_zipper_res4.ply"), mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    scene.animate_light(sc, m["trackball"]["last_time"])
    flat = abi.FlatScene(sc, abi.default_config(640, 480, depth=3), cam)
    check_against_oracle(flat)


def _random_scene(seed, with_mesh):
    """Random primitives (every kind, every material type, reflective and transmissive ones among them) in front of a
    randomly placed camera, one to three lights; optionally the small bunny under a random rotation and scale (a tree
    nobody looked at before). Seeded: the same scenes every run."""
    import math
    from ray_tracer_amd.three_f32 import Euler, Vector3 as V
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(rng.uniform(a, b))
    sc = scene.Scene()
    ids = sc.ids

    def nid(name):
        v = getattr(ids, name)
        setattr(ids, name, v + 1)
        return v

    for _ in range(int(rng.integers(1, 4))):
        sc.lights.append(scene.PointLight(nid("light"), V(u(-400, 400), u(50, 400), u(-500, 200)), V(u(0.2, 1), u(0.2, 1), u(0.2, 1))))
    sc.n_lights_declared = ids.light
    sc.ambientLight = V(u(0.2, 0.9), u(0.2, 0.9), u(0.2, 0.9))
    sc.bgColor = V(u(0, 1), u(0, 1), u(0, 1))

    def material():
        kind = int(rng.integers(0, 3))
        kr = u(0.2, 0.8) if rng.random() < 0.7 else 0.0
        kt = u(0.3, 0.9) if rng.random() < 0.5 else 0.0
        common = (u(0.05, 0.6), u(0.1, 0.9), u(0.1, 1.0), float(rng.integers(4, 40)), kr, kt, u(0.7, 1.6))
        col = lambda: V(u(0, 1), u(0, 1), u(0, 1))
        if kind == 0:
            m = scene.Phong(nid("phong"), *common, col(), col())
        elif kind == 1:
            m = scene.PhongBlinn(nid("blinn"), *common, col(), col())
        else:
            m = scene.PhongCheckered(nid("checkered"), *common, col(), col(), col())
        sc.materials.append(m)
        return m

    kinds = [scene.Sphere, scene.Cube, scene.Cylinder, scene.Rectangle, scene.Sphere, scene.Cube, scene.Cylinder]
    rng.shuffle(kinds)
    for cls in kinds[:int(rng.integers(3, 8))]:
        o = cls(nid("object"))
        if cls is scene.Rectangle:
            o.scale(V(u(100, 600), 1, u(100, 900)))
            o.rotate(Euler(u(-0.3, 0.3), u(-1, 1), u(-0.3, 0.3), "XYZ"))
            o.translate(V(u(-100, 100), u(-120, -40), u(-400, -200)))
        else:
            o.scale(V(u(15, 70), u(15, 70), u(15, 70)))
            o.rotate(Euler(u(-math.pi, math.pi), u(-math.pi, math.pi), u(-math.pi, math.pi), "XYZ"))
            o.translate(V(u(-150, 150), u(-60, 120), u(-450, -150)))
        o.setMaterial(material())
        sc.objects.append(o)
    if rng.random() < 0.5:   # an infinite plane far below
        pl = scene.Plane(nid("object"))
        pl.translate(V(0, u(-300, -150), 0))
        pl.setMaterial(material())
        sc.objects.append(pl)
    if with_mesh:
        v, f = scene.load_ply(model("bun_zipper_res4.ply"))
        mesh = scene.Mesh(nid("object"), v, f, sc.meshes)
        s = u(500, 1200)
        mesh.scale(V(s, s, s))
        mesh.rotate(Euler(u(-1, 1), u(-math.pi, math.pi), u(-1, 1), "XYZ"))
        mesh.translate(V(u(-60, 60), u(-40, 60), u(-320, -200)))
        mesh.setMaterial(material())
        sc.objects.append(mesh)
    # camera: somewhere in front of the scene, looking at its middle; corner rays of a 45 degree frustum in float32
    eye = np.array([u(-120, 120), u(-20, 150), u(-60, 120)])
    fwd = np.array([u(-40, 40), u(-20, 40), -270.0]) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0.0, 1.0, 0.0])
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    th = math.tan(math.radians(22.5))
    cam = {"cameraPos": eye.astype(np.float32)}
    for name, (x, y) in {"ray00": (-1, -1), "ray01": (-1, 1), "ray10": (1, -1), "ray11": (1, 1)}.items():
        d = fwd + right * (x * th * 1.5) + up * (y * th)
        cam[name] = (d / np.linalg.norm(d)).astype(np.float32)
    return sc, cam


def test_repeated_frames_are_bit_identical():
    """Which wave walks which ray, who steals what and the order of the atomics change from run to run; the frame must
    not: twelve traces of one scene (full bunny, depth 5, 960x540) in one context give the same float colours, bytes
    and hit ids every time."""
    flat = make_flat("bun_zipper.ply", 960, 540, depth=5, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    rt = abi.RayTracerHIP(flat.cfg)
    rt.setScene(flat, kd_on_device=True)
    first = None
    for k in range(12):
        rt.trace()
        got = (sha(rt.readFloat()), sha(rt.readRGBA8()), sha(rt.readHitIds()), rt.stats().rays_total)
        if first is None:
            first = got
        assert got == first, k
    rt.close()


def test_axis_parallel_rays_and_an_unrotated_cube_off_to_the_side():
    """The cube's slab test divides by the direction components: a ray with an exactly zero component (the centre column
    and row of an odd-sized frame) sees NaN bounds on that axis, which min / max drop - the cube is then unbounded along
    it, and the centre column reports hits on a cube that stands 100 units to the side. The shader's arithmetic, kept by
    the oracle; the kernels' bounding-sphere filter must not remove those hits."""
    from ray_tracer_amd.three_f32 import Vector3 as V
    sc = scene.Scene()
    ids = sc.ids
    sc.lights.append(scene.PointLight(0, V(50, 200, 100), V(1, 1, 1)))
    ids.light = 1
    sc.n_lights_declared = 1
    m = scene.Phong(0, 0.3, 0.6, 0.4, 16, 0.3, 0.0, 1.0, V(0.8, 0.2, 0.2), V(1, 1, 1))
    ids.phong = 1
    sc.materials.append(m)
    for k, (x, y) in enumerate(((100, 0), (0, 90), (-120, -60))):
        cube = scene.Cube(k)
        cube.scale(V(40, 40, 40))
        cube.translate(V(x, y, -300))
        cube.setMaterial(m)
        sc.objects.append(cube)
    ids.object = 3
    flat = abi.FlatScene(sc, abi.default_config(121, 81, depth=3), scene.camera_rays(121, 81))
    rt, o, (_, _, got_id), cn = check_against_oracle(flat)
    assert (got_id[:, 60] == 0).any()      # the centre column "hits" cube 0 although it stands to the right of it
    assert (got_id[40, :] == 1).any()      # and the centre row the cube above it
    rt.close()


@pytest.mark.parametrize("seed", list(range(1, 25)))
def test_random_scenes_match_the_oracle(seed):
    """Scenes nobody tuned: random transforms (rotated cylinders and cubes, stretched spheres), random materials of the
    three types, random lights and camera; every other scene holds the mesh, every fifth is supersampled, the recursion
    depth runs through 2 .. 6, both tone operators. Everything `check_against_oracle` compares must be equal."""
    with_mesh, ss = seed % 2 == 1, 2 if seed % 5 == 0 else 1
    sc, cam = _random_scene(seed, with_mesh)
    flat = abi.FlatScene(sc, abi.default_config(120, 80, depth=2 + seed % 5, supersampling=ss, tr_method=seed & 1), cam)
    rt, o, (_, got_8, _), cn = check_against_oracle(flat)
    assert cn.rays >= 120 * 80 * ss * ss
    Lbar, sf = rt.tone()
    oL, osf = o.tone(got_8)
    assert abs(Lbar - oL) <= TONE_RTOL * abs(oL) and abs(sf - osf) <= 1e-8 * abs(osf)
    assert np.array_equal(rt.present(), o.present(got_8, oL, osf))
    rt.close()


def test_kdtree_export_through_gpu_context():
    m = manifest("k8_res3_640x480")
    flat = make_flat("bun_zipper_res3.ply", 64, 48)
    rt = abi.RayTracerHIP(flat.cfg)
    rt.setScene(flat)
    pairs, leaves, box = rt.exportKdTree()
    assert sha(pairs) == m["kd"]["sha256"]["nodes_sparse_u32"] and sha(leaves) == m["kd"]["sha256"]["leaves_i32"]
    assert sha(box) == m["kd"]["sha256"]["root_box_f64"]


def test_empty_mesh_and_no_primitives():
    flat = make_flat(None, 64, 48, depth=2)
    flat.prims = []
    rt = render_hip(flat)
    assert (rt.readHitIds() == -1).all()
    assert (rt.readRGBA8()[..., :3] == np.array([26, 128, 230], np.uint8)).all()
    assert rt.stats().rays_total == 64 * 48 and rt.stats().shadow_rays == 0


@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_sharded_render_assembles_to_the_full_frame(n_shards):
    """n shard contexts on one GPU (n = 3 does not divide the tile count; the frame is ragged on both edges): tiles
    interleave, partial tone sums add up, the assembled render target equals the unsharded one byte for byte - and so
    does the canvas when every shard tone-maps its own tiles (rt_present_shard) and the root only scatters them."""
    sharded_render_check(n_shards, 203, 117, 3)


def sharded_render_check(n_shards, width, height, depth):
    import ctypes as C
    flat = make_flat("bun_zipper_res4.ply", width, height, depth=depth)
    full = render_hip(flat)
    want = full.readRGBA8()
    ws, wn = full.tonePartial()
    full.toneFinish(ws, wn)
    want_canvas = full.present()
    L = abi.lib()
    hip = C.CDLL("libamdhip64.so")

    def fetch(fn, rt):
        p, nb = C.c_void_p(), C.c_size_t()
        assert fn(rt.ctx, C.byref(p), C.byref(nb)) == 0
        host = np.zeros(nb.value, np.uint8)
        if nb.value:
            assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), p, C.c_size_t(nb.value), 2) == 0
        return host

    def upload(bufs):
        stride = -(-max(len(b) for b in bufs) // 256) * 256
        gathered = np.zeros(stride * len(bufs), np.uint8)
        for r, b in enumerate(bufs):
            gathered[r * stride:r * stride + len(b)] = b
        dptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(dptr), C.c_size_t(len(gathered))) == 0
        assert hip.hipMemcpy(dptr, C.c_void_p(gathered.ctypes.data), C.c_size_t(len(gathered)), 1) == 0
        return dptr, stride

    parts, sums, cnt, bufs = [], 0.0, 0, []
    for r in range(n_shards):
        flat.cfg.shard_index, flat.cfg.shard_count = r, n_shards
        rt = render_hip(flat)
        s, n = rt.tonePartial()
        sums += s
        cnt += n
        bufs.append(fetch(L.rt_device_shard_rgba8, rt))
        parts.append(rt)
    flat.cfg.shard_index, flat.cfg.shard_count = 0, 1
    assert cnt == wn == width * height
    assert abs(sums - ws) <= 1e-12 * abs(ws)
    root = abi.RayTracerHIP(flat.cfg)
    dptr, stride = upload(bufs)
    assert L.rt_assemble_shards(root.ctx, dptr, C.c_size_t(stride), n_shards) == 0
    assert np.array_equal(root.readRGBA8(), want)
    hip.hipFree(dptr)
    # the canvas by shards: every shard finishes the tone values with the SAME global sums, presents its own tiles
    canv = []
    for rt in parts:
        rt.toneFinish(ws, wn)
        rt.presentShard()
        canv.append(fetch(L.rt_device_shard_canvas, rt))
    dptr, stride = upload(canv)
    assert L.rt_assemble_shards_canvas(root.ctx, dptr, C.c_size_t(stride), n_shards) == 0
    assert np.array_equal(device_canvas(root), want_canvas)
    hip.hipFree(dptr)
    # a stride that no shard of this frame can have is refused
    assert L.rt_assemble_shards(root.ctx, dptr, C.c_size_t(stride * 64), n_shards) != 0
    for rt in parts + [root, full]:
        rt.close()


def test_config3_as_benchmarked_bunny_and_teapot_whole_4k_frame():
    """The scene bench.py times (bench.build_workload: full bunny + teapot.obj with rebased indices, reflective and
    transmissive mesh, 3840x2160, depth 5, two lights), compared with the oracle over the WHOLE frame: every primary
    hit id, every RGBA8 byte, every float colour, every per-level ray count. The teapot stands behind the camera
    (js/6-transmission.js:2010-2014 places it there), so it is seen only through reflections: secondary rays that hit
    its faces are counted to make sure the comparison covers them."""
    import bench
    sc = bench.build_workload()
    cfg = abi.default_config(3840, 2160, depth=5)
    from ray_tracer_amd import scene
    flat = abi.FlatScene(sc, cfg, scene.camera_rays(3840, 2160))
    assert len(flat.faces) == 69451 + 2464
    rt, o, (gf, g8, gid), cn = check_against_oracle(flat)
    st = rt.stats()
    assert st.pixels == 3840 * 2160 and st.rays_total > 2 * st.pixels
    assert st.kd_closest_rays == sum(cn.rays_in_box) and st.kd_shadow_rays == sum(cn.shadow_in_box)
    # the teapot's box lies at z > -172 + ..., i.e. its faces are hit by rays that travel towards +z: reflections
    n_bunny = 69451
    assert (gid < len(flat.prims) + n_bunny).all()        # no primary ray sees the teapot
    # pixels whose colour differs from the same frame without the teapot = pixels that see it in a reflection
    flat2 = abi.FlatScene(bench.build_workload(teapot=False), abi.default_config(3840, 2160, depth=5), scene.camera_rays(3840, 2160))
    no_teapot = render_hip(flat2).readRGBA8()
    changed = (no_teapot != g8).any(axis=-1)
    assert changed.sum() > 5000, changed.sum()
    Lbar, sf = rt.tone()
    wl, ws = o.tone(g8)
    assert abs(Lbar - wl) <= 1e-8 * wl and np.float32(sf) == np.float32(ws)
    assert np.array_equal(rt.present(), o.present(g8, wl, ws))


def test_full_size_config3_properties_and_sampled_parity():
    """benchmark configuration 3 without the second mesh (69k-triangle bunny, depth 5, 3840x2160): parity with the oracle on
    four 96x64 windows (sky+floor, sphere, bunny) and on a lattice of every third pixel in x and y over the whole frame,
    and frame-level properties: idempotence (two traces give the same bytes), ray accounting, background pixels exact."""
    flat = make_flat("bun_zipper.ply", 3840, 2160, depth=5, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    rt = render_hip(flat)
    a8, af, aid = rt.readRGBA8(), rt.readFloat(), rt.readHitIds()
    st = rt.stats()
    assert st.pixels == 3840 * 2160 and st.rays_per_level[0] == st.pixels
    assert st.rays_total == sum(st.rays_per_level)
    for l in range(1, 5):
        assert st.rays_per_level[l] <= 2 * st.rays_per_level[l - 1]
    miss = aid == -1
    assert miss.any() and (a8[miss][:, :3] == np.array([26, 128, 230], np.uint8)).all()
    rt.trace()
    assert np.array_equal(rt.readRGBA8(), a8)
    o = oracle.OracleRenderer(flat)
    for rect in ((3700, 2050, 3796, 2114), (1900, 1000, 1996, 1064), (2000, 1900, 2096, 1964), (600, 300, 696, 364)):
        wf, w8, wid, _ = o.trace(rect=rect, threads=16)
        sl = (slice(rect[1], rect[3]), slice(rect[0], rect[2]))
        assert np.array_equal(aid[sl], wid[sl]), rect
        assert np.array_equal(a8[sl], w8[sl]), rect
        assert ((af[sl] == wf[sl]) | (np.isnan(af[sl]) & np.isnan(wf[sl]))).all(), rect
    wf, w8, wid, _ = o.trace(step=3, threads=16)
    lat = (slice(0, None, 3), slice(0, None, 3))
    assert np.array_equal(aid[lat], wid[lat]) and np.array_equal(a8[lat], w8[lat])
    assert ((af[lat] == wf[lat]) | (np.isnan(af[lat]) & np.isnan(wf[lat]))).all()
    Lbar, sf = rt.tone()
    wl, ws = o.tone(a8)
    assert abs(Lbar - wl) <= 1e-8 * wl and abs(sf - ws) <= 1e-8 * ws      # 8.3 M sequential adds in the reference
    assert np.float32(sf) == np.float32(ws)                                 # what the present shader receives


def test_parts_do_not_change_the_image():
    """rt_config.parts traces the shard as concurrent slices on separate streams; every output must be the same."""
    outs = []
    for parts in (1, 3):
        flat = make_flat("bun_zipper_res4.ply", 331, 203, depth=4, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9), parts=parts)
        rt = abi.RayTracerHIP(flat.cfg)
        rt.setScene(flat)
        rt.trace()
        st = rt.stats()
        outs.append((sha(rt.readFloat()), sha(rt.readRGBA8()), sha(rt.readHitIds()), rt.tone(), sha(rt.present()),
                     st.rays_total, st.shadow_rays, list(st.rays_per_level)))
        rt.close()
    assert outs[0] == outs[1]


def test_pipelined_frames_equal_synchronous_ones():
    """Two contexts alternate frames with everything queued on their streams (rt_trace_begin, device-resident tone,
    queued present, rt_trace_end one frame later): every frame must equal the one-call-at-a-time result. The light
    moves between frames, so a context that mixed up frames would show."""
    import ctypes as C
    import torch
    from ray_tracer_amd import multi_gpu, scene
    flat = make_flat("bun_zipper_res4.ply", 331, 203, depth=4, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    rts = []
    for _ in range(2):
        rt = abi.RayTracerHIP(flat.cfg)
        rt.setScene(flat)
        rt.setOption("walk_waves_per_cu", 12)
        rts.append(rt)
    comm = multi_gpu.Comm(0, 1, 0, backend=None, device="cuda:0")
    job = multi_gpu.PipelinedFrameJob(rts, comm)
    ref = abi.RayTracerHIP(flat.cfg)
    ref.setScene(flat)

    def lights_at(t):
        x = 500.0 * np.cos(2.0 * np.pi * (t % 60.0) / 60.0)
        return [((x, 300.0, x - 280.0), flat.lights[0][1])] + list(flat.lights[1:])

    times = [0.0, 3.0, 6.0, 9.0, 12.0]
    for i, t in enumerate(times):
        rts[i % 2].setLights(lights_at(t), flat.n_lights_loop)
        job.step()
    job.flush()
    torch.cuda.synchronize()
    for k, t in ((0, times[4]), (1, times[3])):      # the last frame each context traced
        ref.setLights(lights_at(t), flat.n_lights_loop)
        ref.trace()
        wl, ws = ref.tone()
        want_canvas = ref.present()
        rt = rts[k]
        assert np.array_equal(rt.readRGBA8(), ref.readRGBA8())
        assert np.array_equal(rt.readFloat(), ref.readFloat(), equal_nan=True)
        gl, gs = rt.toneGet()
        assert abs(gl - wl) <= 1e-12 * wl and abs(gs - ws) <= 1e-12 * ws
        assert np.float32(gs) == np.float32(ws)
        p, n = C.c_void_p(), C.c_size_t()
        assert rt.L.rt_device_canvas(rt.ctx, C.byref(p), C.byref(n)) == 0
        canvas = torch.as_tensor(multi_gpu._DevView(p.value, n.value), device="cuda:0").cpu().numpy()
        assert np.array_equal(canvas.reshape(want_canvas.shape), want_canvas)
        assert rt.stats().rays_total == ref.stats().rays_total


def test_pipelined_job_with_device_collectives_one_rank():
    """The N > 1 branch of the pipelined job on a one-rank RCCL group: the all-reduce runs in place on the library's
    device memory, the gather fills the root's buffer, rt_assemble_shards scatters it back - all queued on the
    context's stream. With one rank the exchange must leave the frame unchanged."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from ray_tracer_amd import multi_gpu
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    flat = make_flat("bun_zipper_res4.ply", 203, 117, depth=3, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    rts = []
    for _ in range(2):
        rt = abi.RayTracerHIP(flat.cfg)
        rt.setScene(flat)
        rts.append(rt)
    comm = multi_gpu.Comm(0, 1, 0, backend="nccl", device="cuda:0", single_rank_group=True)
    try:
        ref = abi.RayTracerHIP(flat.cfg)
        ref.setScene(flat)
        ref.trace()
        wl, ws = ref.tone()
        want_canvas = ref.present().reshape(flat.cfg.height, flat.cfg.width, 4)
        # "root": the root assembles the render target and presents it; "shards": every rank presents its own tiles
        # (rt_present_shard) and the root assembles the canvas
        for mode in ("root", "shards"):
            job = multi_gpu.PipelinedFrameJob(rts, comm, present=mode)
            assert job.by_shards == (mode == "shards")
            for rt in rts:   # no canvas left over from the other mode
                p, n = abi.C.c_void_p(), abi.C.c_size_t()
                rt._ck(rt.L.rt_device_canvas(rt.ctx, abi.C.byref(p), abi.C.byref(n)), "rt_device_canvas")
                torch.as_tensor(multi_gpu._DevView(p.value, n.value), device="cuda:0").zero_()
            torch.cuda.synchronize()
            for _ in range(4):
                job.step()
            job.flush()
            torch.cuda.synchronize()
            for rt in rts:
                assert np.array_equal(rt.readRGBA8(), ref.readRGBA8())
                gl, gs = rt.toneGet()
                assert abs(gl - wl) <= 1e-12 * wl and np.float32(gs) == np.float32(ws)
                assert np.array_equal(device_canvas(rt), want_canvas), mode
            assert job.redone == 0 and job.joined == 1 and job.finished == 4
        # contexts whose buffers are far too small for their first frame: the trace raises its flag on the device, the
        # flag travels through the all-reduce, and the exchange of those frames is repeated (no host-side collective)
        small = []
        for _ in range(2):
            rt = abi.RayTracerHIP(flat.cfg)
            rt.setScene(flat)
            rt.setOption("first_level_rays", 256)
            small.append(rt)
        job = multi_gpu.PipelinedFrameJob(small, comm, present="shards")
        for _ in range(4):
            job.step()
        job.flush()
        torch.cuda.synchronize()
        assert job.redone == 2
        job.close()
        for rt in small:
            assert np.array_equal(rt.readRGBA8(), ref.readRGBA8())
            assert np.array_equal(device_canvas(rt), want_canvas)
            rt.close()
    finally:
        comm.close()


def test_supersampled_4k_target_ward_and_reinhard_present():
    """A 1920x1080 canvas with SUPERSAMPLING 2 (3840x2160 render target, as the project scene is configured):
    traced windows equal the oracle, and the present pass - 2x2 box filter, Ward and Reinhard operators - equals the
    oracle's over the whole canvas."""
    flat = make_flat("bun_zipper_res2.ply", 1920, 1080, depth=2, ss=2, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    rt = render_hip(flat)
    a8, aid = rt.readRGBA8(), rt.readHitIds()
    assert a8.shape == (2160, 3840, 4)
    o = oracle.OracleRenderer(flat)
    for rect in ((1900, 1000, 1996, 1064), (2000, 1900, 2096, 1964)):
        _, w8, wid, _ = o.trace(rect=rect, threads=16, want_float=False)
        sl = (slice(rect[1], rect[3]), slice(rect[0], rect[2]))
        assert np.array_equal(aid[sl], wid[sl]) and np.array_equal(a8[sl], w8[sl]), rect
    Lbar, sf = rt.tone()
    wl, ws = o.tone(a8)
    assert abs(Lbar - wl) <= 1e-8 * wl and np.float32(sf) == np.float32(ws)
    ward = rt.present()
    assert ward.shape == (1080, 1920, 4)
    assert np.array_equal(ward, o.present(a8, wl, ws))
    flat.cfg.tr_method = 1
    rt2 = render_hip(flat)
    rt2.tone()
    reinhard = rt2.present()
    assert np.array_equal(reinhard, oracle.OracleRenderer(flat).present(a8, wl, ws))
    assert not np.array_equal(reinhard, ward)


def test_reinhard_and_ward_present_onto_a_3840x2160_canvas():
    """benchmark configuration 4 at its size: the 3840x2160 frame of the bunny scene (depth 2) presented with both tone operators;
    Lbar / sf against the reference's sequential loop, and the whole canvas equal to the oracle's present pass."""
    for method in (1, 0):
        flat = make_flat("bun_zipper_res2.ply", 3840, 2160, depth=2, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9), tr_method=method)
        rt = render_hip(flat)
        g8 = rt.readRGBA8()
        o = oracle.OracleRenderer(flat)
        Lbar, sf = rt.tone()
        wl, ws = o.tone(g8)
        assert abs(Lbar - wl) <= 1e-8 * wl and np.float32(sf) == np.float32(ws) and np.float32(Lbar) == np.float32(wl)
        canvas = rt.present()
        assert canvas.shape == (2160, 3840, 4)
        assert np.array_equal(canvas, o.present(g8, wl, ws)), method
        rt.close()


def test_buffers_too_small_for_the_first_frame_are_grown_and_the_frame_repeated():
    """rt_trace sizes the ray-tree levels and the leaf-record buffers from earlier frames; a frame that does not fit
    is traced again with larger buffers (rt_trace_end returns 1). Started far too small, the result must still be
    exact, and the following frame must fit without a repeat."""
    flat = make_flat("bun_zipper_res4.ply", 331, 203, depth=4, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    want = render_hip(flat)
    rt = abi.RayTracerHIP(flat.cfg)
    rt.setScene(flat)
    rt.setOption("first_level_rays", 256)
    rt.traceBegin()
    assert rt.traceEnd() is True
    assert np.array_equal(rt.readRGBA8(), want.readRGBA8())
    assert np.array_equal(rt.readFloat(), want.readFloat(), equal_nan=True)
    assert np.array_equal(rt.readHitIds(), want.readHitIds())
    assert rt.stats().rays_total == want.stats().rays_total and rt.stats().shadow_rays == want.stats().shadow_rays
    rt.traceBegin()
    assert rt.traceEnd() is False
    assert np.array_equal(rt.readRGBA8(), want.readRGBA8())


def test_pipelined_job_repeats_the_exchange_of_a_retraced_frame():
    """Contexts that start with buffers far too small: rt_trace_end reports the repeat one frame later, and the job
    queues the tone reduction and the present pass of that frame again. The frames must come out exact."""
    import ctypes as C
    import torch
    from ray_tracer_amd import multi_gpu
    flat = make_flat("bun_zipper_res4.ply", 203, 117, depth=3, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    ref = render_hip(flat)
    wl, ws = ref.tone()
    want_canvas = ref.present()
    rts = []
    for _ in range(2):
        rt = abi.RayTracerHIP(flat.cfg)
        rt.setScene(flat)
        rt.setOption("first_level_rays", 256)
        rts.append(rt)
    job = multi_gpu.PipelinedFrameJob(rts, multi_gpu.Comm(0, 1, 0, backend=None, device="cuda:0"))
    job.step()
    job.step()
    job.flush()
    torch.cuda.synchronize()
    assert job.redone == 2
    for rt in rts:
        assert np.array_equal(rt.readRGBA8(), ref.readRGBA8())
        gl, gs = rt.toneGet()
        assert abs(gl - wl) <= 1e-12 * wl and np.float32(gs) == np.float32(ws)
        p, n = C.c_void_p(), C.c_size_t()
        assert rt.L.rt_device_canvas(rt.ctx, C.byref(p), C.byref(n)) == 0
        canvas = torch.as_tensor(multi_gpu._DevView(p.value, n.value), device="cuda:0").cpu().numpy()
        assert np.array_equal(canvas.reshape(want_canvas.shape), want_canvas)
