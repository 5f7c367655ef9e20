'use strict';
// The few three.js r84 types the scene-description API of the reference touches (js/8-kd_tree.js:1094-1098,
// :1180-1181, :1245-1257, :2055-2096, :2674, :2934-3022), with r84's mixed precision: Matrix4 elements live in a
// Float32Array (every store rounds to single precision), Vector3 components and all arithmetic are doubles.
// Vector2, Quaternion, EventDispatcher and the camera's position / quaternion / lookAt pose are what three.js's own
// examples/js/controls/TrackballControls.js needs (the reference's UI creates one, js/8-kd_tree.js:2680-2688): that
// file runs unchanged against this module.
// No WebGL, no DOM: this is all of "THREE" that the render path needs.

class Vector2 {
    constructor(x, y) {
        this.x = x || 0;
        this.y = y || 0;
    }
    set(x, y) { this.x = x; this.y = y; return this; }
    clone() { return new Vector2(this.x, this.y); }
    copy(v) { this.x = v.x; this.y = v.y; return this; }
    add(v) { this.x += v.x; this.y += v.y; return this; }
    sub(v) { this.x -= v.x; this.y -= v.y; return this; }
    subVectors(a, b) { this.x = a.x - b.x; this.y = a.y - b.y; return this; }
    multiplyScalar(s) {
        if (isFinite(s)) { this.x *= s; this.y *= s; } else { this.x = 0; this.y = 0; }
        return this;
    }
    lengthSq() { return this.x * this.x + this.y * this.y; }
    length() { return Math.sqrt(this.x * this.x + this.y * this.y); }
    distanceToSquared(v) { const dx = this.x - v.x, dy = this.y - v.y; return dx * dx + dy * dy; }
    equals(v) { return v.x === this.x && v.y === this.y; }
}

class Quaternion {
    constructor(x, y, z, w) {
        this.x = x || 0;
        this.y = y || 0;
        this.z = z || 0;
        this.w = w !== undefined ? w : 1;
    }
    set(x, y, z, w) { this.x = x; this.y = y; this.z = z; this.w = w; return this; }
    clone() { return new Quaternion(this.x, this.y, this.z, this.w); }
    copy(q) { this.x = q.x; this.y = q.y; this.z = q.z; this.w = q.w; return this; }
    // axis must be normalised
    setFromAxisAngle(axis, angle) {
        const half = angle / 2, s = Math.sin(half);
        this.x = axis.x * s;
        this.y = axis.y * s;
        this.z = axis.z * s;
        this.w = Math.cos(half);
        return this;
    }
    // the upper 3x3 of m must be a pure rotation; branch order and expressions as in r84
    setFromRotationMatrix(m) {
        const te = m.elements;
        const m11 = te[0], m12 = te[4], m13 = te[8], m21 = te[1], m22 = te[5], m23 = te[9], m31 = te[2], m32 = te[6], m33 = te[10];
        const trace = m11 + m22 + m33;
        let s;
        if (trace > 0) {
            s = 0.5 / Math.sqrt(trace + 1.0);
            this.w = 0.25 / s;
            this.x = (m32 - m23) * s;
            this.y = (m13 - m31) * s;
            this.z = (m21 - m12) * s;
        } else if (m11 > m22 && m11 > m33) {
            s = 2.0 * Math.sqrt(1.0 + m11 - m22 - m33);
            this.w = (m32 - m23) / s;
            this.x = 0.25 * s;
            this.y = (m12 + m21) / s;
            this.z = (m13 + m31) / s;
        } else if (m22 > m33) {
            s = 2.0 * Math.sqrt(1.0 + m22 - m11 - m33);
            this.w = (m13 - m31) / s;
            this.x = (m12 + m21) / s;
            this.y = 0.25 * s;
            this.z = (m23 + m32) / s;
        } else {
            s = 2.0 * Math.sqrt(1.0 + m33 - m11 - m22);
            this.w = (m21 - m12) / s;
            this.x = (m13 + m31) / s;
            this.y = (m23 + m32) / s;
            this.z = 0.25 * s;
        }
        return this;
    }
}

class EventDispatcher {
    addEventListener(type, listener) {
        if (this._listeners === undefined) this._listeners = {};
        const l = this._listeners;
        if (l[type] === undefined) l[type] = [];
        if (l[type].indexOf(listener) === -1) l[type].push(listener);
    }
    hasEventListener(type, listener) {
        return this._listeners !== undefined && this._listeners[type] !== undefined && this._listeners[type].indexOf(listener) !== -1;
    }
    removeEventListener(type, listener) {
        if (this._listeners === undefined || this._listeners[type] === undefined) return;
        const i = this._listeners[type].indexOf(listener);
        if (i !== -1) this._listeners[type].splice(i, 1);
    }
    dispatchEvent(event) {
        if (this._listeners === undefined || this._listeners[event.type] === undefined) return;
        event.target = this;
        for (const f of this._listeners[event.type].slice()) f.call(this, event);
    }
}

class Vector3 {
    constructor(x, y, z) {
        this.x = x || 0;
        this.y = y || 0;
        this.z = z || 0;
    }
    set(x, y, z) { this.x = x; this.y = y; this.z = z; return this; }
    clone() { return new Vector3(this.x, this.y, this.z); }
    copy(v) { this.x = v.x; this.y = v.y; this.z = v.z; return this; }
    add(v) { this.x += v.x; this.y += v.y; this.z += v.z; return this; }
    sub(v) { this.x -= v.x; this.y -= v.y; this.z -= v.z; return this; }
    subVectors(a, b) { this.x = a.x - b.x; this.y = a.y - b.y; this.z = a.z - b.z; return this; }
    addVectors(a, b) { this.x = a.x + b.x; this.y = a.y + b.y; this.z = a.z + b.z; return this; }
    // r84 zeroes the vector when the scalar is not finite
    multiplyScalar(s) {
        if (isFinite(s)) { this.x *= s; this.y *= s; this.z *= s; } else { this.x = 0; this.y = 0; this.z = 0; }
        return this;
    }
    // r84 divides by multiplying with the reciprocal
    divideScalar(s) { return this.multiplyScalar(1 / s); }
    divide(v) { this.x /= v.x; this.y /= v.y; this.z /= v.z; return this; }
    setFromMatrixPosition(m) { const e = m.elements; this.x = e[12]; this.y = e[13]; this.z = e[14]; return this; }
    dot(v) { return this.x * v.x + this.y * v.y + this.z * v.z; }
    lengthSq() { return this.x * this.x + this.y * this.y + this.z * this.z; }
    length() { return Math.sqrt(this.x * this.x + this.y * this.y + this.z * this.z); }
    normalize() { return this.divideScalar(this.length()); }
    // in place: this = this x v
    cross(v) {
        const ax = this.x, ay = this.y, az = this.z;
        this.x = ay * v.z - az * v.y;
        this.y = az * v.x - ax * v.z;
        this.z = ax * v.y - ay * v.x;
        return this;
    }
    crossVectors(a, b) {
        const ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
        this.x = ay * bz - az * by;
        this.y = az * bx - ax * bz;
        this.z = ax * by - ay * bx;
        return this;
    }
    min(v) { this.x = Math.min(this.x, v.x); this.y = Math.min(this.y, v.y); this.z = Math.min(this.z, v.z); return this; }
    max(v) { this.x = Math.max(this.x, v.x); this.y = Math.max(this.y, v.y); this.z = Math.max(this.z, v.z); return this; }
    // point transform with perspective divide (through divideScalar, i.e. a reciprocal multiply)
    applyMatrix4(m) {
        const px = this.x, py = this.y, pz = this.z, e = m.elements;
        this.x = e[0] * px + e[4] * py + e[8] * pz + e[12];
        this.y = e[1] * px + e[5] * py + e[9] * pz + e[13];
        this.z = e[2] * px + e[6] * py + e[10] * pz + e[14];
        return this.divideScalar(e[3] * px + e[7] * py + e[11] * pz + e[15]);
    }
    transformDirection(m) {
        const px = this.x, py = this.y, pz = this.z, e = m.elements;
        this.x = e[0] * px + e[4] * py + e[8] * pz;
        this.y = e[1] * px + e[5] * py + e[9] * pz;
        this.z = e[2] * px + e[6] * py + e[10] * pz;
        return this.normalize();
    }
    unproject(camera) {
        const m = new Matrix4().multiplyMatrices(camera.matrixWorld, new Matrix4().getInverse(camera.projectionMatrix));
        return this.applyMatrix4(m);
    }
    equals(v) { return v.x === this.x && v.y === this.y && v.z === this.z; }
    setLength(l) { return this.multiplyScalar(l / this.length()); }
    distanceToSquared(v) { const dx = this.x - v.x, dy = this.y - v.y, dz = this.z - v.z; return dx * dx + dy * dy + dz * dz; }
    // q * v * q^-1, products in r84's order
    applyQuaternion(q) {
        const x = this.x, y = this.y, z = this.z, qx = q.x, qy = q.y, qz = q.z, qw = q.w;
        const ix = qw * x + qy * z - qz * y;
        const iy = qw * y + qz * x - qx * z;
        const iz = qw * z + qx * y - qy * x;
        const iw = -qx * x - qy * y - qz * z;
        this.x = ix * qw + iw * -qx + iy * -qz - iz * -qy;
        this.y = iy * qw + iw * -qy + iz * -qx - ix * -qz;
        this.z = iz * qw + iw * -qz + ix * -qy - iy * -qx;
        return this;
    }
    toArray(out, offset) {
        out = out || [];
        offset = offset || 0;
        out[offset] = this.x; out[offset + 1] = this.y; out[offset + 2] = this.z;
        return out;
    }
}

class Euler {
    constructor(x, y, z, order) {
        this.x = x || 0;
        this.y = y || 0;
        this.z = z || 0;
        this.order = order || 'XYZ';
    }
}

class Matrix4 {
    constructor() {
        this.elements = new Float32Array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]);
    }
    _put(v) { this.elements.set(v); return this; }
    identity() { return this._put([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]); }
    clone() { const m = new Matrix4(); m.elements.set(this.elements); return m; }
    copy(m) { this.elements.set(m.elements); return this; }
    makeTranslation(x, y, z) { return this._put([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, x, y, z, 1]); }
    makeScale(x, y, z) { return this._put([x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1]); }
    makeRotationFromEuler(eu) {
        if (eu.order !== 'XYZ') throw new Error("Matrix4.makeRotationFromEuler: only the 'XYZ' order is implemented");
        const cx = Math.cos(eu.x), sx = Math.sin(eu.x);
        const cy = Math.cos(eu.y), sy = Math.sin(eu.y);
        const cz = Math.cos(eu.z), sz = Math.sin(eu.z);
        const cxcz = cx * cz, cxsz = cx * sz, sxcz = sx * cz, sxsz = sx * sz;
        const r = new Array(16).fill(0);
        r[0] = cy * cz;
        r[4] = -cy * sz;
        r[8] = sy;
        r[1] = cxsz + sxcz * sy;
        r[5] = cxcz - sxsz * sy;
        r[9] = -sx * cy;
        r[2] = sxsz - cxcz * sy;
        r[6] = sxcz + cxsz * sy;
        r[10] = cx * cy;
        r[15] = 1;
        return this._put(r);
    }
    makeRotationFromQuaternion(q) {
        const x = q.x, y = q.y, z = q.z, w = q.w;
        const x2 = x + x, y2 = y + y, z2 = z + z;
        const xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2, wx = w * x2, wy = w * y2, wz = w * z2;
        return this._put([1 - (yy + zz), xy + wz, xz - wy, 0, xy - wz, 1 - (xx + zz), yz + wx, 0, xz + wy, yz - wx, 1 - (xx + yy), 0, 0, 0, 0, 1]);
    }
    // rotation part only (the other elements stay): z = eye - target, x = up x z, y = z x x
    lookAt(eye, target, up) {
        const z = new Vector3().subVectors(eye, target).normalize();
        if (z.lengthSq() === 0) z.z = 1;
        const x = new Vector3().crossVectors(up, z).normalize();
        if (x.lengthSq() === 0) {
            z.z += 0.0001;
            x.crossVectors(up, z).normalize();
        }
        const y = new Vector3().crossVectors(z, x);
        const te = this.elements;
        te[0] = x.x; te[4] = y.x; te[8] = z.x;
        te[1] = x.y; te[5] = y.y; te[9] = z.y;
        te[2] = x.z; te[6] = y.z; te[10] = z.z;
        return this;
    }
    scale(v) {
        const te = this.elements, x = v.x, y = v.y, z = v.z;
        te[0] *= x; te[4] *= y; te[8] *= z;
        te[1] *= x; te[5] *= y; te[9] *= z;
        te[2] *= x; te[6] *= y; te[10] *= z;
        te[3] *= x; te[7] *= y; te[11] *= z;
        return this;
    }
    setPosition(v) { const te = this.elements; te[12] = v.x; te[13] = v.y; te[14] = v.z; return this; }
    compose(position, quaternion, scale) { return this.makeRotationFromQuaternion(quaternion).scale(scale).setPosition(position); }
    multiply(m) { return this.multiplyMatrices(this, m); }
    premultiply(m) { return this.multiplyMatrices(m, this); }
    multiplyMatrices(A, B) {
        const a = A.elements, b = B.elements, r = new Array(16);
        for (let col = 0; col < 4; col++) {
            const b0 = b[4 * col], b1 = b[4 * col + 1], b2 = b[4 * col + 2], b3 = b[4 * col + 3];
            for (let row = 0; row < 4; row++) {
                r[4 * col + row] = a[row] * b0 + a[4 + row] * b1 + a[8 + row] * b2 + a[12 + row] * b3;
            }
        }
        return this._put(r);
    }
    // adjugate / determinant, cofactor products in r84's order
    getInverse(src, throwOnDegenerate) {
        const s = src.elements;
        const n11 = s[0], n21 = s[1], n31 = s[2], n41 = s[3];
        const n12 = s[4], n22 = s[5], n32 = s[6], n42 = s[7];
        const n13 = s[8], n23 = s[9], n33 = s[10], n43 = s[11];
        const n14 = s[12], n24 = s[13], n34 = s[14], n44 = s[15];
        const t11 = n23 * n34 * n42 - n24 * n33 * n42 + n24 * n32 * n43 - n22 * n34 * n43 - n23 * n32 * n44 + n22 * n33 * n44;
        const t12 = n14 * n33 * n42 - n13 * n34 * n42 - n14 * n32 * n43 + n12 * n34 * n43 + n13 * n32 * n44 - n12 * n33 * n44;
        const t13 = n13 * n24 * n42 - n14 * n23 * n42 + n14 * n22 * n43 - n12 * n24 * n43 - n13 * n22 * n44 + n12 * n23 * n44;
        const t14 = n14 * n23 * n32 - n13 * n24 * n32 - n14 * n22 * n33 + n12 * n24 * n33 + n13 * n22 * n34 - n12 * n23 * n34;
        const det = n11 * t11 + n21 * t12 + n31 * t13 + n41 * t14;
        if (det === 0) {
            const msg = "THREE.Matrix4.getInverse(): can't invert matrix, determinant is 0";
            if (throwOnDegenerate) throw new Error(msg);
            return this.identity();
        }
        const k = 1 / det;
        const r = new Array(16);
        r[0] = t11 * k;
        r[1] = (n24 * n33 * n41 - n23 * n34 * n41 - n24 * n31 * n43 + n21 * n34 * n43 + n23 * n31 * n44 - n21 * n33 * n44) * k;
        r[2] = (n22 * n34 * n41 - n24 * n32 * n41 + n24 * n31 * n42 - n21 * n34 * n42 - n22 * n31 * n44 + n21 * n32 * n44) * k;
        r[3] = (n23 * n32 * n41 - n22 * n33 * n41 - n23 * n31 * n42 + n21 * n33 * n42 + n22 * n31 * n43 - n21 * n32 * n43) * k;
        r[4] = t12 * k;
        r[5] = (n13 * n34 * n41 - n14 * n33 * n41 + n14 * n31 * n43 - n11 * n34 * n43 - n13 * n31 * n44 + n11 * n33 * n44) * k;
        r[6] = (n14 * n32 * n41 - n12 * n34 * n41 - n14 * n31 * n42 + n11 * n34 * n42 + n12 * n31 * n44 - n11 * n32 * n44) * k;
        r[7] = (n12 * n33 * n41 - n13 * n32 * n41 + n13 * n31 * n42 - n11 * n33 * n42 - n12 * n31 * n43 + n11 * n32 * n43) * k;
        r[8] = t13 * k;
        r[9] = (n14 * n23 * n41 - n13 * n24 * n41 - n14 * n21 * n43 + n11 * n24 * n43 + n13 * n21 * n44 - n11 * n23 * n44) * k;
        r[10] = (n12 * n24 * n41 - n14 * n22 * n41 + n14 * n21 * n42 - n11 * n24 * n42 - n12 * n21 * n44 + n11 * n22 * n44) * k;
        r[11] = (n13 * n22 * n41 - n12 * n23 * n41 - n13 * n21 * n42 + n11 * n23 * n42 + n12 * n21 * n43 - n11 * n22 * n43) * k;
        r[12] = t14 * k;
        r[13] = (n13 * n24 * n31 - n14 * n23 * n31 + n14 * n21 * n33 - n11 * n24 * n33 - n13 * n21 * n34 + n11 * n23 * n34) * k;
        r[14] = (n14 * n22 * n31 - n12 * n24 * n31 - n14 * n21 * n32 + n11 * n24 * n32 + n12 * n21 * n34 - n11 * n22 * n34) * k;
        r[15] = (n12 * n23 * n31 - n13 * n22 * n31 + n13 * n21 * n32 - n11 * n23 * n32 - n12 * n21 * n33 + n11 * n22 * n33) * k;
        return this._put(r);
    }
    makePerspective(left, right, top, bottom, near, far) {
        const r = new Array(16).fill(0);
        r[0] = 2 * near / (right - left);
        r[8] = (right + left) / (right - left);
        r[5] = 2 * near / (top - bottom);
        r[9] = (top + bottom) / (top - bottom);
        r[10] = -(far + near) / (far - near);
        r[14] = -2 * far * near / (far - near);
        r[11] = -1;
        return this._put(r);
    }
    toArray(out, offset) {
        out = out || [];
        offset = offset || 0;
        for (let i = 0; i < 16; i++) out[offset + i] = this.elements[i];
        return out;
    }
}

const DEG2RAD = Math.PI / 180;

// The pose is position + quaternion (+ scale) as in three.js's Object3D; updateMatrixWorld() composes matrixWorld from
// them with r84's arithmetic (quaternion -> single-precision rotation matrix). The reference leaves the camera where
// TrackballControls.reset() puts it, at the origin looking down -z (js/8-kd_tree.js:2798-2802): matrixWorld starts as
// the identity.
class PerspectiveCamera {
    constructor(fov, aspect, near, far) {
        this.fov = fov !== undefined ? fov : 50;
        this.aspect = aspect !== undefined ? aspect : 1;
        this.near = near !== undefined ? near : 0.1;
        this.far = far !== undefined ? far : 2000;
        this.zoom = 1;
        this.position = new Vector3(0, 0, 0);
        this.quaternion = new Quaternion();
        this.scale = new Vector3(1, 1, 1);
        this.up = new Vector3(0, 1, 0);
        this.matrix = new Matrix4();
        this.matrixWorld = new Matrix4();
        this.matrixWorldInverse = new Matrix4();
        this.projectionMatrix = new Matrix4();
        this._look = new Matrix4();   // r84 keeps one scratch matrix for every lookAt: only its rotation part is ever written
        this.updateProjectionMatrix();
    }
    updateProjectionMatrix() {
        const near = this.near;
        const top = near * Math.tan(DEG2RAD * 0.5 * this.fov) / this.zoom;
        const height = 2 * top;
        const width = this.aspect * height;
        const left = -0.5 * width;
        this.projectionMatrix.makePerspective(left, left + width, top, top - height, near, this.far);
    }
    // Camera.lookAt of r84: the orientation only; matrixWorld follows at the next updateMatrixWorld()
    lookAt(target) {
        this._look.lookAt(this.position, target, this.up);
        this.quaternion.setFromRotationMatrix(this._look);
        return this;
    }
    updateMatrix() { this.matrix.compose(this.position, this.quaternion, this.scale); }
    updateMatrixWorld() {
        this.updateMatrix();
        this.matrixWorld.copy(this.matrix);
        this.matrixWorldInverse.getInverse(this.matrixWorld);
    }
}

module.exports = { Vector2, Vector3, Quaternion, Euler, Matrix4, EventDispatcher, PerspectiveCamera, Math: { DEG2RAD } };
