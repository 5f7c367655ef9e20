"""The handful of three.js r84 routines that sit on the ray tracer's host path, restated with the
same mixed precision: Matrix4 elements live in a Float32Array (every store rounds to f32), Vector3
components and all arithmetic are f64 (reference js/three.min.js: Matrix4 `this.elements=new
Float32Array([...])`, Vector3.applyMatrix4, Matrix4.getInverse/multiplyMatrices/makePerspective/
makeRotationFromEuler; call sites js/8-kd_tree.js:1094-1098, :1245-1257, :2055-2096, :2674).

Used by the Python driver (tests, bench). The product host is the JavaScript twin under js/.
"""
import math

import numpy as np


def _f32(x):
    return float(np.float32(x))


class Vector3:
    __slots__ = ("x", "y", "z")

    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(x), float(y), float(z)

    def clone(self):
        return Vector3(self.x, self.y, self.z)

    def sub(self, v):
        self.x -= v.x
        self.y -= v.y
        self.z -= v.z
        return self

    def multiplyScalar(self, s):
        if math.isfinite(s):
            self.x *= s
            self.y *= s
            self.z *= s
        else:
            self.x = self.y = self.z = 0.0
        return self

    def divideScalar(self, s):
        return self.multiplyScalar(1.0 / s if s != 0 else math.copysign(math.inf, s))

    def length(self):
        return math.sqrt(self.x * self.x + self.y * self.y + self.z * self.z)

    def normalize(self):
        return self.divideScalar(self.length())

    def applyMatrix4(self, m):
        b, c, d = self.x, self.y, self.z
        a = m.e
        self.x = a[0] * b + a[4] * c + a[8] * d + a[12]
        self.y = a[1] * b + a[5] * c + a[9] * d + a[13]
        self.z = a[2] * b + a[6] * c + a[10] * d + a[14]
        return self.divideScalar(a[3] * b + a[7] * c + a[11] * d + a[15])

    def unproject(self, camera):
        m = Matrix4().multiplyMatrices(camera.matrixWorld, Matrix4().getInverse(camera.projectionMatrix))
        return self.applyMatrix4(m)

    def toArray(self):
        return [self.x, self.y, self.z]


class Euler:
    def __init__(self, x=0.0, y=0.0, z=0.0, order="XYZ"):
        if order != "XYZ":
            raise ValueError("only the 'XYZ' order is used by the reference scenes")
        self.x, self.y, self.z, self.order = x, y, z, order


class Matrix4:
    """`e` is a Python list of 16 floats that are always f32-representable (column-major)."""

    __slots__ = ("e",)

    def __init__(self):
        self.e = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0]

    def _store(self, vals):
        self.e = [float(v) for v in np.asarray(vals, dtype=np.float64).astype(np.float32)]
        return self

    def identity(self):
        return self._store([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1])

    def clone(self):
        m = Matrix4()
        m.e = list(self.e)
        return m

    def makeTranslation(self, x, y, z):
        return self._store([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, x, y, z, 1])

    def makeScale(self, x, y, z):
        return self._store([x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1])

    def makeRotationFromEuler(self, eu):
        c, d, e = eu.x, eu.y, eu.z
        f, c = math.cos(c), math.sin(c)
        g, d = math.cos(d), math.sin(d)
        h, e = math.cos(e), math.sin(e)
        a = f * h
        m = f * e
        k = c * h
        l = c * e
        b = [0.0] * 16
        b[0] = g * h
        b[4] = -g * e
        b[8] = d
        b[1] = m + k * d
        b[5] = a - l * d
        b[9] = -c * g
        b[2] = l - a * d
        b[6] = k + m * d
        b[10] = f * g
        b[15] = 1.0
        return self._store(b)

    def multiplyMatrices(self, A, B):
        c, d = A.e, B.e
        f, g, h, m = c[0], c[4], c[8], c[12]
        k, l, p, n = c[1], c[5], c[9], c[13]
        u, q, r, A_ = c[2], c[6], c[10], c[14]
        w, y, K, c15 = c[3], c[7], c[11], c[15]
        v, E, L, C = d[0], d[4], d[8], d[12]
        F, x, H, D = d[1], d[5], d[9], d[13]
        z, J, I, Q = d[2], d[6], d[10], d[14]
        M, O, P, d15 = d[3], d[7], d[11], d[15]
        e = [0.0] * 16
        e[0] = f * v + g * F + h * z + m * M
        e[4] = f * E + g * x + h * J + m * O
        e[8] = f * L + g * H + h * I + m * P
        e[12] = f * C + g * D + h * Q + m * d15
        e[1] = k * v + l * F + p * z + n * M
        e[5] = k * E + l * x + p * J + n * O
        e[9] = k * L + l * H + p * I + n * P
        e[13] = k * C + l * D + p * Q + n * d15
        e[2] = u * v + q * F + r * z + A_ * M
        e[6] = u * E + q * x + r * J + A_ * O
        e[10] = u * L + q * H + r * I + A_ * P
        e[14] = u * C + q * D + r * Q + A_ * d15
        e[3] = w * v + y * F + K * z + c15 * M
        e[7] = w * E + y * x + K * J + c15 * O
        e[11] = w * L + y * H + K * I + c15 * P
        e[15] = w * C + y * D + K * Q + c15 * d15
        return self._store(e)

    def multiply(self, m):
        return self.multiplyMatrices(self, m)

    def getInverse(self, src, throwOnDegenerate=False):
        d = src.e
        e, f, g, h = d[0], d[1], d[2], d[3]
        m, k, l, p = d[4], d[5], d[6], d[7]
        n, u, q, r = d[8], d[9], d[10], d[11]
        A, w, y, dd = d[12], d[13], d[14], d[15]
        K = u * y * p - w * q * p + w * l * r - k * y * r - u * l * dd + k * q * dd
        v = A * q * p - n * y * p - A * l * r + m * y * r + n * l * dd - m * q * dd
        E = n * w * p - A * u * p + A * k * r - m * w * r - n * k * dd + m * u * dd
        L = A * u * l - n * w * l - A * k * q + m * w * q + n * k * y - m * u * y
        C = e * K + f * v + g * E + h * L
        if C == 0:
            if throwOnDegenerate:
                raise ValueError("THREE.Matrix4.getInverse(): can't invert matrix, determinant is 0")
            return self.identity()
        C = 1 / C
        c = [0.0] * 16
        c[0] = K * C
        c[1] = (w * q * h - u * y * h - w * g * r + f * y * r + u * g * dd - f * q * dd) * C
        c[2] = (k * y * h - w * l * h + w * g * p - f * y * p - k * g * dd + f * l * dd) * C
        c[3] = (u * l * h - k * q * h - u * g * p + f * q * p + k * g * r - f * l * r) * C
        c[4] = v * C
        c[5] = (n * y * h - A * q * h + A * g * r - e * y * r - n * g * dd + e * q * dd) * C
        c[6] = (A * l * h - m * y * h - A * g * p + e * y * p + m * g * dd - e * l * dd) * C
        c[7] = (m * q * h - n * l * h + n * g * p - e * q * p - m * g * r + e * l * r) * C
        c[8] = E * C
        c[9] = (A * u * h - n * w * h - A * f * r + e * w * r + n * f * dd - e * u * dd) * C
        c[10] = (m * w * h - A * k * h + A * f * p - e * w * p - m * f * dd + e * k * dd) * C
        c[11] = (n * k * h - m * u * h - n * f * p + e * u * p + m * f * r - e * k * r) * C
        c[12] = L * C
        c[13] = (n * w * g - A * u * g + A * f * q - e * w * q - n * f * y + e * u * y) * C
        c[14] = (A * k * g - m * w * g - A * f * l + e * w * l + m * f * y - e * k * y) * C
        c[15] = (m * u * g - n * k * g + n * f * l - e * u * l - m * f * q + e * k * q) * C
        return self._store(c)

    def makePerspective(self, left, right, top, bottom, near, far):
        a, b, c, d, e, f = left, right, top, bottom, near, far
        g = [0.0] * 16
        g[0] = 2 * e / (b - a)
        g[8] = (b + a) / (b - a)
        g[5] = 2 * e / (c - d)
        g[9] = (c + d) / (c - d)
        g[10] = -(f + e) / (f - e)
        g[14] = -2 * f * e / (f - e)
        g[11] = -1.0
        return self._store(g)

    def toArray(self):
        return list(self.e)


DEG2RAD = math.pi / 180


class PerspectiveCamera:
    """Pose is given directly as matrixWorld; the reference leaves the camera at the origin looking
    down -z (TrackballControls with no input, js/8-kd_tree.js:2798-2802), i.e. the identity."""

    def __init__(self, fov, aspect, near, far):
        self.fov, self.aspect, self.near, self.far, self.zoom = fov, aspect, near, far, 1
        self.position = Vector3(0, 0, 0)
        self.matrixWorld = Matrix4()
        self.projectionMatrix = Matrix4()
        self.updateProjectionMatrix()

    def updateProjectionMatrix(self):
        a = self.near
        b = a * math.tan(0.5 * DEG2RAD * self.fov) / self.zoom
        c = 2 * b
        d = self.aspect * c
        e = -0.5 * d
        self.projectionMatrix.makePerspective(e, e + d, b, b - c, a, self.far)


def get_primary_ray(camera, x, y):
    """getPrimaryRay, js/8-kd_tree.js:1094-1098."""
    v = Vector3(x, y, -1).unproject(camera)
    return v.sub(camera.position).normalize()
