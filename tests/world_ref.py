"""An independent float64 answer for the analytic part of the default scene, used to pin the pixel oracle.

The reference's shader (and the oracle, which restates it) intersects every primitive in OBJECT space through the
inverse model matrix. This module never does: it works in WORLD space from the scene constants of
js/8-kd_tree.js:2927-3024 - spheres by centre and radius, the floor as the plane y = -80 with its bounds, the cube as an
oriented box, the cylinder as an axis, a radius and two cap planes - and applies the shading formulas of
js/8-kd_tree.js:755-813 in double precision. Agreement with the oracle therefore checks the restatement's geometry
code and its illuminate(), not just its self-consistency.
"""
import math

import numpy as np

EPS = 1e-4
SPHERE_EPS = 0.095
RECT_EPS = 0.095


def rot_xyz(ax, ay, az):
    """THREE.Euler(ax, ay, az, 'XYZ') as a rotation matrix: Rx * Ry * Rz."""
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def unit(v):
    return v / np.linalg.norm(v)


class World:
    """The five analytic objects of generateScene() in scene order, the two lights at time t, the materials."""

    def __init__(self, time=0.0):
        c = 500.0 * math.cos((time % 60.0) / 60.0 * math.pi * 2.0)
        self.lights = [(np.array([c, 300.0, c - 280.0]), np.array([1.0, 1.0, 1.0])),
                       (np.array([-300.0, 50.0, 200.0]), np.array([0.1, 1.0, 0.1]))]
        self.ambient = np.array([0.75, 0.75, 0.75])
        self.bg = np.array([0.1, 0.5, 0.9])
        # (kind, ka, kd, ks, ke, kReflect, kRefract, ior, Co / (Co1, Co2), Cs)
        one = np.ones(3)
        self.mats = [("phong", 0.075, 0.075, 0.2, 20, 0.01, 0.8, 0.95, one, one),
                     ("phong", 0.15, 0.25, 1.0, 20, 0.75, 0.0, 0.0, 0.7 * one, one),
                     ("checker", 0.5, 0.9, 0.2, 12, 0.4, 0.0, 0.0, (np.array([1.0, 0, 0]), np.array([1.0, 1.0, 0])), 0.7 * one),
                     ("phong", 0.5, 0.9, 0.2, 24, 0.0, 0.0, 0.0, np.array([0.8, 0.1, 0.7]), 0.7 * one),
                     ("phong", 0.5, 0.3, 0.9, 20, 0.0, 0.0, 0.0, np.array([0.85, 0.7, 0.1]), 0.7 * one)]
        self.s1 = (np.array([2.0, 15.0, -240.0]), 50.0)
        self.s2 = (np.array([-75.0, -20.0, -325.0]), 40.0)
        self.floor_c, self.floor_half = np.array([-105.0, -80.0, -300.0]), (225.0, 800.0)
        self.cube_c, self.cube_R, self.cube_half = np.array([-70.0, 40.0, -400.0]), rot_xyz(-math.pi / 4, math.pi / 4, 0.0), np.array([45.0, 30.0, 30.0])
        self.cyl_c, self.cyl_R, self.cyl_r, self.cyl_h = np.array([50.0, 75.0, -300.0]), rot_xyz(-math.pi / 3, 0.0, -math.pi / 6), 30.0, 30.0

    # every test returns None or (t, normal as the shader holds it - NOT normalised -, object-space point or None)
    def sphere(self, which, o, d):
        c, r = which
        oc = o - c
        a, b, cc = d @ d, 2.0 * (oc @ d), oc @ oc - r * r
        disc = b * b - 4 * a * cc
        if disc <= 0:
            return None
        e = math.sqrt(disc)
        t = (-b - e) / (2 * a)
        if t > SPHERE_EPS:
            return t, (o + t * d - c) / r, None               # near root: the unit object-space normal
        t = (-b + e) / (2 * a)
        if t > SPHERE_EPS:
            return t, (o + t * d - c), None                   # far root: model * normal = r * unit normal
        return None

    def floor(self, o, d):
        if d[1] == 0:
            return None
        t = (self.floor_c[1] - o[1]) / d[1]
        if not t > RECT_EPS:
            return None
        p = o + t * d
        u, v = (p[0] - self.floor_c[0]) / 450.0, (p[2] - self.floor_c[2]) / 1600.0      # object-space x, z in (-0.5, 0.5)
        # the reference measures from the corner (-0.5, 0, 0.5) along +x and -z: both coordinates strictly inside (0, 1)
        if 0.0 < u + 0.5 < 1.0 and 0.0 < 0.5 - v < 1.0:
            return t, np.array([0.0, 1.0, 0.0]), np.array([u, 0.0, v])
        return None

    def cube(self, o, d):
        R, h = self.cube_R, self.cube_half
        lo, ld = R.T @ (o - self.cube_c), R.T @ d              # box frame, world units
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = -h / np.abs(ld) - lo / ld
            t1 = h / np.abs(ld) - lo / ld
        tn, tf = t0.max(), t1.min()
        if tn > tf or tf < 0:
            return None
        k = int(np.argmax(t0))
        n = np.zeros(3)
        n[k] = -np.sign(ld[k])
        return tn, R @ (n * 2.0 * h), None                     # model * n: the axis scaled by the full edge length

    def cylinder(self, o, d):
        R, r, h = self.cyl_R, self.cyl_r, self.cyl_h
        a = R @ np.array([0.0, 1.0, 0.0])
        oc = o - self.cyl_c
        dp, op = d - (d @ a) * a, oc - (oc @ a) * a           # components perpendicular to the axis
        A, B, C = dp @ dp, 2.0 * (op @ dp), op @ op - r * r
        disc = B * B - 4 * A * C
        if disc <= 0:
            return None
        e = math.sqrt(disc)
        t0, t1 = sorted(((-B - e) / (2 * A), (-B + e) / (2 * A)))
        y0, y1 = (oc + t0 * d) @ a, (oc + t1 * d) @ a          # heights along the axis, caps at +-h
        if y0 < -h:
            if y1 >= -h:
                th = t0 + (t1 - t0) * (y0 + h) / (y0 - y1)
                if th > 0:
                    return th, -a * 2.0 * h, None
        elif y0 <= h:
            if t0 > 0:
                q = oc + t0 * d
                return t0, (q - (q @ a) * a), None             # r * radial unit vector = model * (lx, 0, lz)
        else:
            if y1 <= h:
                th = t0 + (t1 - t0) * (y0 - h) / (y0 - y1)
                if th > 0:
                    return th, a * 2.0 * h, None
        return None

    def tests(self, o, d):
        return [self.sphere(self.s1, o, d), self.sphere(self.s2, o, d), self.floor(o, d), self.cube(o, d), self.cylinder(o, d)]

    def closest(self, o, d):
        best = None
        for i, h in enumerate(self.tests(o, d)):
            if h is not None and (best is None or h[0] < best[1][0]):     # strict <: the earlier object keeps a tie
                best = (i, h)
        return best

    def occluded(self, o, d):
        return any(h is not None for h in self.tests(o, d))

    def shade(self, o, d):
        """Local colour of the closest hit of ray (o, d), as illuminate() computes it; None on a miss."""
        hit = self.closest(o, d)
        if hit is None:
            return None
        i, (t, n_raw, local) = hit
        kind, ka, kd, ks, ke, _, _, _, Co, Cs = self.mats[i]
        P = o + t * d
        if kind == "checker":
            row, col = int((local[0] + 0.5) * 12.0), int((local[2] + 0.5) / (450.0 / 19200.0))
            Co = Co[0] if (row % 2 == 0) == (col % 2 == 0) else Co[1]
        N, V = unit(n_raw), unit(-d)
        colour = ka * self.ambient * Co
        blocked = 0
        for Lp, Lc in self.lights:
            S = unit(Lp - P)
            if self.occluded(P + n_raw * EPS, S):
                blocked += 1
                continue
            sn = max(0.0, S @ N)
            colour = colour + min(max(kd * sn, 0.0), 1.0) * Lc * Co
            if sn > 0 and kind == "phong":
                Rv = unit(-S - 2.0 * (N @ -S) * N)
                colour = colour + min(max(ks * max(0.0, Rv @ V) ** ke, 0.0), 1.0) * Lc * Cs
        return {"id": i, "t": t, "point": P, "normal": n_raw, "colour": colour, "blocked": blocked, "local": local}


def primary_dir(px, py, W, H, fov_deg=45.0):
    """Direction of the primary ray through the centre of pixel (px, py), py = 0 at the bottom: the four corner rays are
    equally long, so their bilinear mix followed by a normalisation is the normalised pinhole direction."""
    th = math.tan(math.radians(fov_deg) / 2.0)
    x = ((px + 0.5) / W * 2.0 - 1.0) * th * (W / H)
    y = ((py + 0.5) / H * 2.0 - 1.0) * th
    return unit(np.array([x, y, -1.0]))
