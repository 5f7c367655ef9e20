"""Known answers for the pixel half of the oracle (the reference GLSL cannot be executed here, so these
are derived by hand from the scene: js/8-kd_tree.js:2927-3024 and the shader text :327-1057)."""
import math

import numpy as np

from helpers import make_flat
from oracle import oracle


def test_pow_accuracy_and_edges():
    L = oracle.lib()
    assert L.oracle_pow(0.0, 20.0) == 0.0 and L.oracle_pow(1.0, 20.0) == 1.0
    rng = np.random.default_rng(3)
    worst = 0.0
    for x, y in zip(rng.uniform(0.0, 1.0, 4000), rng.choice([12.0, 16.0, 20.0, 24.0, 3.5, 100.0], 4000)):
        xf = float(np.float32(x))
        want = np.float32(math.pow(xf, y))
        got = np.float32(L.oracle_pow(xf, y))
        if want > 1e-30:
            worst = max(worst, abs(float(got) - float(want)) / float(np.spacing(want)))
    assert worst <= 1.0, worst     # within one float32 ulp of the correctly rounded power


def test_background_pixel_is_bgcolor():
    flat = make_flat(None, 64, 48, depth=5)
    o = oracle.OracleRenderer(flat)
    rgb, rgba, ids, cn = o.trace(threads=1)
    # the top-right corner looks at the sky: no object, no floor
    assert ids[47, 63] == -1
    assert np.array_equal(rgb[47, 63], np.array([0.1, 0.5, 0.9], np.float32))
    assert list(rgba[47, 63]) == [26, 128, 230, 255]          # round(c*255)
    assert cn.rays >= 64 * 48


def test_centre_pixel_hits_sphere1_and_spawns_both_children():
    # the four centre pixels of a 640x480 frame: primary rays within 0.1 degree of (0,0,-1), which hits
    # sphere1 (centre (2,15,-240), radius 50)
    flat = make_flat(None, 640, 480, depth=2)
    o = oracle.OracleRenderer(flat)
    rgb, rgba, ids, cn = o.trace(rect=(319, 239, 321, 241), threads=1)
    assert (ids[239:241, 319:321] == 0).all()
    # sphere1 has kReflect 0.01 and kRefract 0.8: every hit spawns a reflected and a refracted ray
    assert cn.rays_level[0] == 4 and cn.rays_level[1] == 8
    # two lights, one shadow ray per light per shaded node (all 12 nodes hit something or the sky)
    assert cn.shadow_rays % 2 == 0


def test_floor_checker_and_ambient_term():
    # depth 1 image; a floor pixel in full shadow of nothing: colour >= ambient term ka*ambient*Co
    flat = make_flat(None, 64, 48, depth=1)
    o = oracle.OracleRenderer(flat)
    rgb, rgba, ids, cn = o.trace(threads=1)
    floor = ids == 2
    assert floor.sum() > 200
    # checker colours are Co1 = (1,0,0) and Co2 = (1,1,0): blue stays 0 wherever the specular term is 0,
    # red >= 0.5*0.75*1 everywhere on the floor
    assert (rgb[floor][:, 0] >= 0.375 - 1e-6).all()
    greens = rgb[floor][:, 1]
    assert (greens < 1e-6).any() and (greens >= 0.375 - 1e-6).any()      # both checker colours present
    # specularCoeff of the checkered material stays 0 (no branch assigns it): blue is exactly 0
    assert (rgb[floor][:, 2] == 0.0).all()


def test_depth1_has_exactly_one_ray_per_pixel():
    flat = make_flat("bun_zipper_res4.ply", 80, 60, depth=1)
    o = oracle.OracleRenderer(flat)
    _, _, ids, cn = o.trace(threads=2)
    assert cn.rays == 80 * 60 and cn.rays_level[1] == 0
    assert (ids >= 5).sum() > 100          # the bunny is in view (ids >= n_prims are faces)
    assert cn.rays_in_box[0] > 0 and cn.kd_steps[0] > cn.kd_leaf_visits[0] > 0


def test_restart_traversal_equals_brute_force_over_listed_faces():
    """With use_kd_tree = 0 the shader loops over all faces. The k-d result may differ only through the
    four-face leaf truncation (a hole), never by reporting a farther triangle than brute force finds nearer
    among the faces the tree still lists."""
    kd = make_flat("bun_zipper_res4.ply", 96, 72, depth=1)
    bf = make_flat("bun_zipper_res4.ply", 96, 72, depth=1, use_kd_tree=0)
    a = oracle.OracleRenderer(kd).trace(threads=2)
    b = oracle.OracleRenderer(bf).trace(threads=2)
    same = a[2] == b[2]
    assert same.mean() > 0.97
    # where they differ, brute force saw a mesh triangle that the truncated tree lost or replaced
    assert (b[2][~same] >= 5).all()


def test_present_ward_and_reinhard():
    flat = make_flat(None, 32, 24, depth=1, ss=1)
    o = oracle.OracleRenderer(flat)
    _, rgba, _, _ = o.trace(threads=1)
    Lbar, sf = o.tone(rgba)
    canvas = o.present(rgba, Lbar, sf)
    # Ward: out = byte/255 * L_MAX * sf / L_DMAX = 2 * sf * byte/255, quantised
    want = np.clip(np.float32(rgba[..., :3]) / np.float32(255) * np.float32(1000) * np.float32(sf) / np.float32(500), 0, 1)
    want8 = (want * np.float32(255) + np.float32(0.5)).astype(np.int32)
    assert np.abs(canvas[..., :3].astype(int) - want8).max() <= 0
    flat.cfg.tr_method = 1
    o2 = oracle.OracleRenderer(flat)
    c2 = o2.present(rgba, Lbar, sf)
    x = np.float32(rgba[..., :3]) / np.float32(255) * np.float32(1000) * np.float32(0.18) / np.float32(Lbar)
    w2 = np.clip(x * np.float32(500) / (np.float32(1) + x) / np.float32(500), 0, 1)
    assert np.abs(c2[..., :3].astype(int) - (w2 * 255 + 0.5).astype(int)).max() <= 1


# ---- numeric known answers -----------------------------------------------------------------------------------------
# The pixel oracle restates the shader's object-space code; tests/world_ref.py answers the same questions in world space
# and double precision straight from the scene constants (js/8-kd_tree.js:2927-3024) and the shading formulas
# (js/8-kd_tree.js:755-813). Tolerances: 1e-5 relative on distances and directions (float32 arithmetic against
# float64), 2e-4 absolute on colours (pow(x, 20) multiplies the relative error of x by 20).
import world_ref  # noqa: E402

REL = 1e-5


def _analytic_oracle(w=640, h=480, depth=1):
    flat = make_flat(None, w, h, depth=depth)
    return flat, oracle.OracleRenderer(flat)


def test_world_ref_rotation_convention_matches_the_pinned_model_matrices():
    """The model matrices are pinned bit for bit on the reference's uniforms (tests/test_host_golden.py); the world-space
    description used below must describe the same objects."""
    flat, _ = _analytic_oracle()
    W = world_ref.World()
    for idx, (c, R, scale) in ((3, (W.cube_c, W.cube_R, np.array([90.0, 60.0, 60.0]))), (4, (W.cyl_c, W.cyl_R, np.array([30.0, 60.0, 30.0])))):
        M = np.array(flat.prims[idx][3], np.float64).reshape(4, 4).T
        assert np.allclose(M[:3, :3], R * scale, rtol=0, atol=2e-5) and np.allclose(M[:3, 3], c, atol=1e-5)
    M = np.array(flat.prims[2][3], np.float64).reshape(4, 4).T
    assert np.allclose(np.diag(M)[:3], [450, 1, 1600]) and np.allclose(M[:3, 3], W.floor_c)
    for idx, (c, r) in ((0, W.s1), (1, W.s2)):
        M = np.array(flat.prims[idx][3], np.float64).reshape(4, 4).T
        assert np.allclose(M[:3, :3], np.eye(3) * r) and np.allclose(M[:3, 3], c)


def test_primary_hit_distance_on_sphere1_and_floor_closed_form():
    flat, o = _analytic_oracle()
    cam = np.zeros(3)
    # centre pixel: the ray (almost) along -z meets sphere1 (centre (2,15,-240), radius 50)
    d = world_ref.primary_dir(320, 240, 640, 480)
    c, r = np.array([2.0, 15.0, -240.0]), 50.0
    t_want = d @ c - math.sqrt((d @ c) ** 2 - c @ c + r * r)
    p = o.probe(cam, d)
    assert p["hit"] and p["id"] == 0 and abs(p["t"] - t_want) <= REL * t_want
    assert np.allclose(p["point"], t_want * d, rtol=REL)
    assert np.allclose(p["normal"], (t_want * d - c) / r, atol=1e-5)          # near root: unit object-space normal
    # a pixel low in the frame: the floor plane y = -80, t = -80 / d.y
    d = world_ref.primary_dir(400, 60, 640, 480)
    t_want = -80.0 / d[1]
    P = t_want * d
    assert -330 < P[0] < 120 and -1100 < P[2] < 500
    p = o.probe(cam, d)
    assert p["hit"] and p["id"] == 2 and abs(p["t"] - t_want) <= REL * t_want
    assert np.allclose(p["local"], [(P[0] + 105) / 450, 0.0, (P[2] + 300) / 1600], atol=1e-6)
    assert np.allclose(p["normal"], [0, 1, 0])


def test_refracted_and_reflected_child_rays_closed_form():
    """js/8-kd_tree.js:1010-1036: eta = 1/ior outside, k = 1 + eta^2 (cosi^2 - 1), T = eta*I + (eta*cosi - sqrt(k))*N with
    the shader's un-normalised normal; R = I - 2 (N.I) N. Checked on sphere1 (transmissive, ior 0.95) from outside, from
    inside (the far root, whose normal is 50 times longer) and on sphere2 (mirror)."""
    flat, o = _analytic_oracle()
    d = world_ref.primary_dir(300, 250, 640, 480)
    p = o.probe(np.zeros(3), d)
    assert p["id"] == 0 and abs(p["ior"] - 0.95) < 1e-6 and abs(p["kRefract"] - 0.8) < 1e-6
    N = p["normal"]
    assert d @ N < 0                                                          # outside
    eta = 1.0 / 0.95
    cosi = -N @ d
    k = 1.0 + eta * eta * (cosi * cosi - 1.0)
    T = eta * d + (eta * cosi - math.sqrt(k)) * N
    assert np.allclose(p["refr_d"], T, rtol=REL, atol=1e-6)
    assert np.allclose(p["refr_o"], p["point"] - N * 1e-4, rtol=REL)
    assert np.allclose(p["refl_d"], d - 2.0 * (N @ d) * N, rtol=REL, atol=1e-6)
    # Snell: sin(theta_t) = eta sin(theta_i) for this unit normal
    sin_i = math.sqrt(1 - cosi ** 2)
    sin_t = np.linalg.norm(np.cross(world_ref.unit(T), -N))
    assert abs(sin_t - eta * sin_i) < 1e-5
    # the refracted ray continues inside the sphere and leaves through the far root: normal = model * local = P - c
    p2 = o.probe(p["refr_o"], p["refr_d"])
    c = np.array([2.0, 15.0, -240.0])
    assert p2["id"] == 0 and np.allclose(p2["normal"], p2["point"] - c, atol=2e-3)
    assert abs(np.linalg.norm(p2["point"] - c) - 50.0) < 1e-2
    d2, N2 = p["refr_d"], p2["normal"]
    assert d2 @ N2 > 0                                                        # inside: eta = ior, normal flipped
    Nf = -N2
    cosi2 = -Nf @ d2
    k2 = 1.0 + 0.95 ** 2 * (cosi2 * cosi2 - 1.0)
    if k2 >= 0:
        assert np.allclose(p2["refr_d"], 0.95 * d2 + (0.95 * cosi2 - math.sqrt(k2)) * Nf, rtol=1e-4, atol=1e-3)
    else:
        assert np.isnan(p2["refr_d"]).all()                                   # sqrt of a negative: NaN direction, as in GLSL
    assert np.allclose(p2["refl_o"], p2["point"] - N2 * 1e-4, rtol=REL)       # inside: the reflected ray starts on the inner side
    # sphere2, kReflect 0.75, no transmission
    ds = world_ref.unit(np.array([-75.0, -20.0, -325.0]) + np.array([5.0, 8.0, 0.0]))
    p3 = o.probe(np.zeros(3), ds)
    assert p3["id"] == 1 and abs(p3["kReflect"] - 0.75) < 1e-6 and p3["kRefract"] == 0.0
    assert np.allclose(p3["refl_d"], ds - 2.0 * (p3["normal"] @ ds) * p3["normal"], rtol=REL, atol=1e-6)


def test_cylinder_side_and_cap_hits_world_space():
    flat, o = _analytic_oracle()
    W = world_ref.World()
    a = W.cyl_R @ np.array([0.0, 1.0, 0.0])
    # towards a point of the side surface, from a place level with the cylinder's middle
    side_pt = W.cyl_c + world_ref.unit(np.cross(a, [0.0, 0.0, 1.0])) * 30.0
    org = side_pt + world_ref.unit(side_pt - W.cyl_c) * 100.0 + np.array([0.0, 0.0, 3.0])
    d = world_ref.unit(side_pt - org)
    want, got = W.cylinder(org, d), o.probe(org, d)
    assert got["id"] == 4 and abs(got["t"] - want[0]) <= 1e-4 * want[0]
    assert np.allclose(got["normal"], want[1], atol=2e-3)
    assert abs(np.linalg.norm(got["normal"]) - 30.0) < 1e-2                   # side: model * (lx, 0, lz), radius 30
    # almost down the axis from beyond the top cap (a ray exactly parallel to the axis has a zero discriminant and hits
    # nothing, in the shader as here): the cap plane lies 100 along the axis, normal = model * (0, 1, 0) = 60 * axis
    perp = world_ref.unit(np.cross(a, [1.0, 0.0, 0.0]))
    org = W.cyl_c + a * 130.0 + perp * 4.0
    d = world_ref.unit(-a + 0.05 * perp)
    t_plane = 100.0 / (d @ -a)
    want, got = W.cylinder(org, d), o.probe(org, d)
    assert abs(want[0] - t_plane) < 1e-9 and np.allclose(want[1], a * 60.0)
    assert got["id"] == 4 and abs(got["t"] - t_plane) <= 1e-4 * t_plane and np.allclose(got["normal"], a * 60.0, atol=1e-3)
    # and up from below: the bottom cap, normal -60 * axis
    org = W.cyl_c - a * 40.0 + perp * 4.0          # 10 below the bottom cap, clear of sphere1
    d = world_ref.unit(a + 0.05 * perp)
    got = o.probe(org, d)
    assert W.closest(org, d)[0] == 4
    assert got["id"] == 4 and abs(got["t"] - 10.0 / (d @ a)) <= 1e-3 and np.allclose(got["normal"], -a * 60.0, atol=1e-3)


def test_checker_parity_at_known_points():
    """js/8-kd_tree.js:756-776: row = int((x + 0.5) * 12), col = int((z + 0.5) / (450/19200)); equal parity -> Co1 (red),
    else Co2 (yellow). Object-space (x, z) = ((X + 105) / 450, (Z + 300) / 1600)."""
    flat, o = _analytic_oracle()
    amb = 0.5 * 0.75
    for (lx, lz), yellow in (((-0.5 + 0.5 / 12, -0.5 + 0.5 * 450 / 19200), False),      # row 0, col 0
                             ((-0.5 + 1.5 / 12, -0.5 + 0.5 * 450 / 19200), True),       # row 1, col 0
                             ((-0.5 + 1.5 / 12, -0.5 + 1.5 * 450 / 19200), False),      # row 1, col 1
                             ((-0.5 + 6.5 / 12, -0.5 + 21.5 * 450 / 19200), True)):     # row 6, col 21
        P = np.array([lx * 450 - 105, -80.0, lz * 1600 - 300])
        org = P + np.array([0.3, 15.0, 0.2])          # below sphere2, which hangs over part of the floor
        p = o.probe(org, world_ref.unit(P - org))
        assert p["id"] == 2 and np.allclose(p["local"], [lx, 0, lz], atol=1e-5)
        green = p["colour"][1]
        assert p["colour"][0] >= amb - 1e-6 and p["colour"][2] == 0.0          # red in both colours; no specular, no blue
        assert (green >= amb - 1e-6) == yellow and (green < 1e-6) == (not yellow), (lx, lz, p["colour"])


def test_phong_colours_against_world_space_reference_on_a_lattice():
    """Every 8th pixel of the 640x480 depth-1 frame (4800 rays): hit id, distance, local colour and the number of blocked
    lights from the oracle against the world-space double-precision answer. Pixels within a hair of a silhouette, a
    shadow edge or a checker line may fall on the other side in float32; they are counted, not compared."""
    flat, o = _analytic_oracle()
    W = world_ref.World(0.0)
    rgb, _, ids, _ = o.trace(step=8, threads=4)
    n = n_same = 0
    seen = {}
    worst = 0.0
    for py in range(0, 480, 8):
        for px in range(0, 640, 8):
            d = world_ref.primary_dir(px, py, 640, 480)
            want = W.shade(np.zeros(3), d)
            n += 1
            wid = -1 if want is None else want["id"]
            if wid != ids[py, px]:
                continue
            if want is None:
                assert np.array_equal(rgb[py, px], np.float32([0.1, 0.5, 0.9]))
                n_same += 1
                continue
            p = o.probe(np.zeros(3), d)
            if p["lights_blocked"] != want["blocked"]:
                continue
            err = np.abs(rgb[py, px].astype(np.float64) - want["colour"]).max()
            if err > 2e-4:
                continue
            assert abs(p["t"] - want["t"]) <= 1e-4 * want["t"]
            worst = max(worst, err)
            n_same += 1
            key = (wid, want["blocked"])
            seen[key] = seen.get(key, 0) + 1
    assert n == 4800 and n_same >= 0.985 * n, (n_same, n)
    # the sample covers every object, lit by both lights and with at least one light blocked somewhere
    for obj in range(5):
        assert any(k[0] == obj for k in seen), obj
    assert seen.get((3, 0), 0) > 20 and seen.get((2, 1), 0) + seen.get((2, 2), 0) > 20      # an unshadowed cube pixel, shadowed floor pixels
    assert worst <= 2e-4


def test_generated_shader_constants_match_the_oracle_configuration():
    """What the reference bakes into its GLSL for configs 1-3 (captured from the generated shader text by the fixture
    generator) against what the oracle is configured with by the same scene description."""
    from helpers import manifest
    for gold, kw in (("c1_ch6_640x480", dict(model_name=None, depth=5, time=None, second_light=False)),
                     ("c2_res4_1920x1080", dict(model_name="bun_zipper_res4.ply", depth=1)),
                     ("c3_bunny_teapot", dict(model_name="bun_zipper_res4.ply", depth=5))):
        g = manifest(gold)["glsl_counts"]
        flat = make_flat(**kw)
        assert g["ray_tree_size"] == 2 ** flat.cfg.max_depth - 1
        assert g["light_loop"] == flat.n_lights_loop
        kinds = {"Plane": 0, "Rect": 1, "Sphere": 2, "Cube": 3, "Cylinder": 4}
        assert [kinds[k] for k in g["intersect_order"]] == [p[0] for p in flat.prims]
        assert g["material_consts"] == [[p[1], p[2]] for p in flat.prims]
        c = manifest(gold)["glsl_counts"]["constants"]          # as the generated shader text spells them
        if gold == "c1_ch6_640x480":
            flat.cfg.infinity = 100000.1                          # js/6-transmission.js:22; make_flat's default is the later scripts' value
        for name, have in (("INFINITY", flat.cfg.infinity), ("EPSILON", flat.cfg.epsilon), ("SPHERE_EPSILON", flat.cfg.sphere_epsilon),
                           ("PLANE_RECT_EPSILON", flat.cfg.plane_rect_epsilon), ("TRIANGLE_EPSILON", flat.cfg.triangle_epsilon)):
            if c.get(name) is not None:
                assert np.float32(c[name]) == np.float32(have), (gold, name, c[name], have)
