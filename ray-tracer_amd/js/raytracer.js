'use strict';
// class RayTracer with the reference's four methods (js/8-kd_tree.js:2298-2589): constructor(), setScene(scene),
// update(time), render(). Everything the reference did through WebGL happens in librt_hip.so instead:
//   constructor  render target allocation                         -> rt_create
//   setScene     shader generation + texture allocation (:2347-2431) -> primitive / material tables
//   update       uniform marshalling, vertex/face upload, k-d build, draw, readPixels, Lbar/sf loop (:2433-2573)
//                -> rt_set_* / rt_set_mesh / rt_update_vertices / rt_build_kdtree / rt_trace / rt_tone
//   render       present draw (:2575-2589)                        -> rt_present into this.canvasPixels
// There is no CPU path: if the addon cannot be loaded, requiring this module throws.
const path = require('path');
const addon = require(path.join(__dirname, 'addon', 'rt_addon.node'));
const { PLANE } = require('./scene.js');

const SIZES = addon.structSizes();
const CONFIG_BYTES = 80, PRIM_BYTES = 144, MATERIAL_BYTES = 72, LIGHT_BYTES = 24;
if (SIZES.config !== CONFIG_BYTES || SIZES.prim !== PRIM_BYTES || SIZES.material !== MATERIAL_BYTES || SIZES.light !== LIGHT_BYTES) {
    throw new Error('rt_addon: struct layout differs from the one raytracer.js packs: ' + JSON.stringify(SIZES));
}

function packConfig(G, device, shardIndex, shardCount) {
    const buf = addon.defaultConfig();
    const dv = new DataView(buf);
    const ints = [G.WIDTH | 0, G.HEIGHT | 0, G.SUPERSAMPLING, G.MAX_RECURSION_DEPTH, G.TONE_REPRODUCTION_METHOD, G.RENDER_TRIANGLES ? 1 : 0,
        G.USE_KD_TREE ? 1 : 0, G.MAX_KD_TREE_DEPTH];
    ints.forEach((v, i) => dv.setInt32(4 * i, v, true));
    const floats = [G.L_MAX, G.L_DMAX, G.A, G.INFINITY, G.EPSILON, G.SPHERE_EPSILON, G.PLANE_RECT_EPSILON, G.TRIANGLE_EPSILON];
    floats.forEach((v, i) => dv.setFloat32(32 + 4 * i, v, true));
    dv.setInt32(64, shardIndex, true);
    dv.setInt32(68, shardCount, true);
    dv.setInt32(72, device, true);
    dv.setInt32(76, G.PARTS | 0, true);   // concurrent slices of the shard, 0 = library default
    return buf;
}

function makeRayTracer(G, THREE, api) {
    const vec3 = (v) => [v.x, v.y, v.z];

    class RayTracer {
        constructor() {
            this.width = G.WIDTH | 0;
            this.height = G.HEIGHT | 0;
            const S = G.SUPERSAMPLING;
            this.ctx = addon.create(packConfig(G, G.DEVICE === undefined ? 0 : G.DEVICE, G.SHARD_INDEX || 0, G.SHARD_COUNT || 1));
            this.pixelBytes = this.width * S * this.height * S * 4;
            this.canvasPixels = new Uint8Array(this.width * this.height * 4);
            this.scene = { objects: [], materials: [], lights: [] };
            this.uniforms = {};
            this.startTime = 0;
            this.Lbar = 0;
            this.sf = 0;
            this._meshKey = null;
            this._treeVerts = null;
        }

        dispose() { addon.destroy(this.ctx); this.ctx = null; }

        setScene(scene) {
            this.uniforms = {};
            this.sampleCount = 0;
            this.scene = scene;
            this.geometryUpdatedOnce = false;
            this._meshKey = null;
            this._treeVerts = null;
        }

        // uniform dictionary exactly as the reference fills it (js/8-kd_tree.js:2435-2453)
        _collectUniforms(time) {
            const sc = this.scene;
            for (let i = 0; i < sc.objects.length; i++) sc.objects[i].setUniforms(this);
            for (let i = 0; i < sc.materials.length; i++) sc.materials[i].setUniforms(this);
            for (let i = 0; i < sc.lights.length; i++) sc.lights[i].setUniforms(this);
            this.uniforms.ambientLight = sc.ambientLight;
            this.uniforms.bgColor = sc.bgColor;
            this.uniforms.cameraPos = G.camera.position.clone();
            this.uniforms.ray00 = G.getPrimaryRay(-1, -1);
            this.uniforms.ray01 = G.getPrimaryRay(-1, +1);
            this.uniforms.ray10 = G.getPrimaryRay(+1, -1);
            this.uniforms.ray11 = G.getPrimaryRay(+1, +1);
            this.uniforms.time = time;
        }

        _uploadTables() {
            const sc = this.scene;
            const prims = sc.objects.filter((o) => !(o instanceof api.Mesh));
            const pbuf = new ArrayBuffer(prims.length * PRIM_BYTES);
            const pdv = new DataView(pbuf);
            prims.forEach((o, i) => {
                if (typeof o.kind !== 'number' || o.kind < PLANE) throw new Error('RayTracer: object ' + i + ' is not a known primitive');
                if (!o.material) throw new Error('RayTracer: object ' + i + ' has no material');
                const base = i * PRIM_BYTES;
                pdv.setInt32(base, o.kind, true);
                pdv.setInt32(base + 4, o.material.getType(), true);
                pdv.setInt32(base + 8, o.material.getId(), true);
                const m = o.modelTransform.elements, inv = o.invModelTransform.elements;
                for (let k = 0; k < 16; k++) {
                    pdv.setFloat32(base + 16 + 4 * k, m[k], true);
                    pdv.setFloat32(base + 80 + 4 * k, inv[k], true);
                }
            });
            addon.setPrimitives(this.ctx, pbuf);

            const mbuf = new ArrayBuffer(sc.materials.length * MATERIAL_BYTES);
            const mdv = new DataView(mbuf);
            sc.materials.forEach((mat, i) => {
                const base = i * MATERIAL_BYTES;
                mdv.setInt32(base, mat.getType(), true);
                mdv.setInt32(base + 4, mat.getId(), true);
                [mat.ka, mat.kd, mat.ks, mat.ke, mat.kReflect, mat.kRefract, mat.ior].forEach((v, k) => mdv.setFloat32(base + 8 + 4 * k, v, true));
                const Co = mat.Co1 || mat.Co, Co2 = mat.Co2 || new THREE.Vector3();
                [...vec3(Co), ...vec3(Co2), ...vec3(mat.Cs)].forEach((v, k) => mdv.setFloat32(base + 36 + 4 * k, v, true));
            });
            addon.setMaterials(this.ctx, mbuf);

            // the shader loops over nextPointLightId lights (js/8-kd_tree.js:787); a light that was constructed
            // but never pushed into scene.lights keeps zero uniforms
            const nLoop = Math.max(G.nextPointLightId | 0, sc.lights.reduce((n, l) => Math.max(n, l.id + 1), 0));
            const lbuf = new ArrayBuffer(nLoop * LIGHT_BYTES);
            const ldv = new DataView(lbuf);
            for (const l of sc.lights) {
                [...vec3(l.position), ...vec3(l.color)].forEach((v, k) => ldv.setFloat32(l.id * LIGHT_BYTES + 4 * k, v, true));
            }
            addon.setLights(this.ctx, lbuf, nLoop);
            addon.setGlobals(this.ctx, new Float32Array(vec3(sc.ambientLight)), new Float32Array(vec3(sc.bgColor)));
            const u = this.uniforms;
            addon.setCamera(this.ctx, new Float32Array([...vec3(u.cameraPos), ...vec3(u.ray00), ...vec3(u.ray01), ...vec3(u.ray10), ...vec3(u.ray11)]));
        }

        // g_verts / g_faces -> the device, as RayTracer.update does it (js/8-kd_tree.js:2458-2539): the vertex and face
        // arrays go up on every frame (unless STATIC_GEOMETRY and done once), the tree and its root box are rebuilt only
        // when none was built yet at a time other than 0 (`this.startTime == 0`) or at least KD_REBUILD_INTERVAL
        // seconds (the reference's 0.30; 0 = every frame) have passed since the last build. In between the new
        // vertices are rendered through the OLD tree and box (rt_update_vertices). Arrays that did not change are not
        // sent again, and a rebuild from unchanged arrays is skipped (it would give the same tree).
        _updateGeometry(time) {
            if (!G.RENDER_TRIANGLES) return;
            if (G.STATIC_GEOMETRY && this.geometryUpdatedOnce) return;
            const interval = G.KD_REBUILD_INTERVAL === undefined ? 0.30 : G.KD_REBUILD_INTERVAL;
            const gv = G.g_verts, gf = G.g_faces;
            const k = this._meshKey;
            // vertices: packed for the addon and compared with the last frame's in the same pass
            const verts = new Float64Array(gv.length * 3);
            const kv = k && k.verts.length === verts.length ? k.verts : null;
            let sameVerts = kv !== null;
            for (let i = 0, j = 0; i < gv.length; i++, j += 3) {
                const v = gv[i], x = v.x, y = v.y, z = v.z;
                verts[j] = x; verts[j + 1] = y; verts[j + 2] = z;
                if (sameVerts && (kv[j] !== x || kv[j + 1] !== y || kv[j + 2] !== z)) sameVerts = false;   // (NaN counts as changed)
            }
            // faces and their materials: compared with the arrays of the last frame where they stand; only a frame that
            // finds a difference (or the first one) packs new arrays
            let faces = k ? k.faces : null, mats = k ? k.mats : null;
            let sameFaces = !!k && faces.length === gf.length * 3 && !!kv;
            for (let i = 0; sameFaces && i < gf.length; i++) {
                const f = gf[i], ix = f.indices, m = f.getMaterial();
                if (!m) throw new Error('RayTracer: face ' + i + ' has no material');
                if (faces[3 * i] !== ix[0] || faces[3 * i + 1] !== ix[1] || faces[3 * i + 2] !== ix[2] ||
                    mats[2 * i] !== m.getType() || mats[2 * i + 1] !== m.getId()) sameFaces = false;
            }
            if (!sameFaces) {
                faces = new Uint32Array(gf.length * 3);
                mats = new Uint32Array(gf.length * 2);
                for (let i = 0; i < gf.length; i++) {
                    const f = gf[i];
                    faces[3 * i] = f.indices[0]; faces[3 * i + 1] = f.indices[1]; faces[3 * i + 2] = f.indices[2];
                    const m = f.getMaterial();
                    if (!m) throw new Error('RayTracer: face ' + i + ' has no material');
                    mats[2 * i] = m.getType(); mats[2 * i + 1] = m.getId();
                }
            }
            sameVerts = sameFaces && sameVerts;
            const same = (a, b) => { if (!b || a.length !== b.length) return false; for (let i = 0; i < a.length; i++) if (a[i] !== b[i]) return false; return true; };
            const useTree = G.USE_KD_TREE && gf.length > 0;
            const rebuildDue = useTree && (this.startTime == 0 || time - this.startTime >= interval);
            const build = () => (G.KD_BUILDER === 'device' ? addon.buildKdTreeDevice : addon.buildKdTree)(this.ctx);
            if (!sameFaces) {
                // another face list: the old leaves would name other faces. (The reference would render exactly that
                // until its next rebuild; its own UI never changes the face list after loading.)
                addon.setMesh(this.ctx, verts, faces, mats);
                if (useTree) { build(); this._treeVerts = verts; this.startTime = time; }
            } else if (rebuildDue) {
                if (!same(verts, this._treeVerts)) { addon.setMesh(this.ctx, verts, faces, mats); build(); this._treeVerts = verts; }
                else if (!sameVerts) addon.updateVertices(this.ctx, verts);   // back where the tree was built
                this.startTime = time;
            } else if (!sameVerts) {
                addon.updateVertices(this.ctx, verts);
            }
            this._meshKey = { verts, faces, mats };
            this.geometryUpdatedOnce = true;
        }

        // everything update() does before the trace: usable on a host-only context (DEVICE = -1)
        prepare(time) {
            this._collectUniforms(time);
            this._uploadTables();
            this._updateGeometry(time);
            if (G.RENDER_TRIANGLES && G.USE_KD_TREE && G.g_faces.length > 0) {
                const [, , box] = [0, 0, this.exportKdBox()];
                this.uniforms.kdBoxMin = new THREE.Vector3(box[0], box[1], box[2]);
                this.uniforms.kdBoxMax = new THREE.Vector3(box[3], box[4], box[5]);
            }
        }

        update(time) {
            this.prepare(time);
            addon.trace(this.ctx);
            const t = addon.tone(this.ctx);
            this.Lbar = t[0];
            this.sf = t[1];
        }

        // update() in two halves (rt_trace_begin / rt_trace_end): updateBegin queues the frame on the GPU and returns,
        // so the caller can prepare the next frame of another RayTracer meanwhile; updateEnd waits for it
        updateBegin(time) {
            this.prepare(time);
            addon.traceBegin(this.ctx);
        }
        updateEnd() {
            addon.traceEnd(this.ctx);   // a repeated trace (buffers grew) is complete when this returns
            const t = addon.tone(this.ctx);
            this.Lbar = t[0];
            this.sf = t[1];
        }

        render() {
            addon.present(this.ctx, this.canvasPixels, this.canvasPixels.length);
            return this.canvasPixels;
        }

        // the reference keeps the read-back render target in this.pixels (js/8-kd_tree.js:2313, :2558); reading it
        // costs a device-to-host copy, so it is fetched on demand
        get pixels() {
            const out = new Uint8Array(this.pixelBytes);
            addon.readRGBA8(this.ctx, out, out.length);
            return out;
        }
        readFloat() {
            const out = new Float32Array(this.pixelBytes / 4 * 3);
            addon.readFloat(this.ctx, out, out.byteLength);
            return out;
        }
        readHitIds() {
            const out = new Int32Array(this.pixelBytes / 4);
            addon.readHitIds(this.ctx, out, out.byteLength);
            return out;
        }
        // What the reference's mouse picking computes on the CPU (js/8-kd_tree.js:2718-2760): the object under canvas
        // pixel (x, y), y counted from the top as mouse events do. Returns the scene object (the Mesh for any of its
        // faces) or null.
        pick(x, y) {
            const S = G.SUPERSAMPLING;
            const id = addon.pick(this.ctx, Math.floor(x * S), Math.floor((this.height - 1 - y) * S + (S - 1)));
            if (id < 0) return null;
            const prims = this.scene.objects.filter((o) => !(o instanceof api.Mesh));
            if (id < prims.length) return prims[id];
            const f = id - prims.length;   // index into g_faces: every Mesh owns a contiguous range of it
            return this.scene.objects.find((o) => o instanceof api.Mesh && f >= o.startFaceIndex && f < o.startFaceIndex + o.facesLength) || null;
        }
        exportKdBox() { const box = new Float64Array(6); addon.kdRootBox(this.ctx, box); return box; }
        exportKdTree() {
            const [nn, nl] = addon.kdtreeCounts(this.ctx);
            const nodes = new Uint32Array(nn * 2), leaves = new Int32Array(nl * 4), box = new Float64Array(6);
            addon.exportKdTree(this.ctx, nodes, leaves, box);
            return { nodes, leaves, box };
        }
        stats() { return addon.stats(this.ctx); }
    }
    return RayTracer;
}

module.exports = { makeRayTracer, addon };
