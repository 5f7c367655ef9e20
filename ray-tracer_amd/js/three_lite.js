'use strict';
// The few three.js r84 types the scene-description API of the reference touches (js/8-kd_tree.js:1094-1098,
// :1180-1181, :1245-1257, :2055-2096, :2674, :2934-3022), with r84's mixed precision: Matrix4 elements live in a
// Float32Array (every store rounds to single precision), Vector3 components and all arithmetic are doubles.
// No WebGL, no DOM: this is all of "THREE" that the render path needs.

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

// Pose = matrixWorld. The reference leaves the camera where TrackballControls.reset() puts it: at the origin,
// looking down -z (js/8-kd_tree.js:2798-2802), so matrixWorld starts as the identity; callers that move the
// camera set position and matrixWorld themselves (lookAt below).
class PerspectiveCamera {
    constructor(fov, aspect, near, far) {
        this.fov = fov !== undefined ? fov : 50;
        this.aspect = aspect !== undefined ? aspect : 1;
        this.near = near !== undefined ? near : 0.1;
        this.far = far !== undefined ? far : 2000;
        this.zoom = 1;
        this.position = new Vector3(0, 0, 0);
        this.up = new Vector3(0, 1, 0);
        this.matrixWorld = new Matrix4();
        this.matrixWorldInverse = new Matrix4();
        this.projectionMatrix = new Matrix4();
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
    // camera at `position` looking at `target`: rotation columns (x, y, z) with z pointing backwards, then the
    // translation; matrixWorldInverse follows
    lookAt(target) {
        const z = new Vector3().subVectors(this.position, target).normalize();
        if (z.lengthSq() === 0) z.z = 1;
        let x = new Vector3().crossVectors(this.up, z).normalize();
        if (x.lengthSq() === 0) { z.z += 0.0001; x = new Vector3().crossVectors(this.up, z).normalize(); }
        const y = new Vector3().crossVectors(z, x);
        this.matrixWorld._put([x.x, x.y, x.z, 0, y.x, y.y, y.z, 0, z.x, z.y, z.z, 0, this.position.x, this.position.y, this.position.z, 1]);
        this.matrixWorldInverse.getInverse(this.matrixWorld);
        return this;
    }
    updateMatrixWorld() { this.matrixWorldInverse.getInverse(this.matrixWorld); }
}

module.exports = { Vector3, Euler, Matrix4, PerspectiveCamera, Math: { DEG2RAD } };
