'use strict';
// The scene-description API of the reference (js/8-kd_tree.js:1162-2291): same class names, constructor
// signatures, methods and error messages, so that an unchanged generateScene() builds its scene with them.
// What is gone is everything that generated GLSL: the classes only carry data for RayTracer to marshal.
//
// The reference keeps its state in page globals (g_verts, g_faces, the next*Id counters, the constants at the
// top of the script). Here they are properties of one object `G` - the global object of the context the scene
// script runs in - so `g_verts = []` inside generateScene() and Mesh.pushToGlobal() meet in the same place.

const PLANE = 0, RECT = 1, SPHERE = 2, CUBE = 3, CYLINDER = 4;   // rt_hip.h primitive kinds

function makeSceneAPI(G, THREE) {
    class Ray {
        constructor(origin, dir) {
            this.origin = origin;
            this.dir = dir;
        }
    }

    // js/8-kd_tree.js:1174-1260
    class Primitive {
        constructor(id) {
            if (this.constructor === Primitive) {
                throw new Error("Primitive: cannot instantiate abstract class.");
            }
            this.modelTransform = new THREE.Matrix4().identity();
            this.invModelTransform = new THREE.Matrix4().identity();
            this.material = null;
            this.id = id;
            this.modelStr = 'model' + id;
            this.invModelStr = 'invModel' + id;
        }
        setMaterial(material) { this.material = material; }
        // the uniform names of js/8-kd_tree.js:1228-1231; RayTracer.update reads them back by primitive
        setUniforms(renderer) {
            renderer.uniforms[this.modelStr] = this.modelTransform;
            renderer.uniforms[this.invModelStr] = this.invModelTransform;
        }
        getMinCorner() { throw new Error("Primitive: cannot call abstract method 'getMinCorner'"); }
        getMaxCorner() { throw new Error("Primitive: cannot call abstract method 'getMaxCorner'"); }
        intersect(ray) { throw new Error("Primitive: cannot call abstract method 'intersect'"); }
        translate(translation) {
            this.modelTransform = new THREE.Matrix4().makeTranslation(translation.x, translation.y, translation.z).multiply(this.modelTransform);
            this.invModelTransform = new THREE.Matrix4().getInverse(this.modelTransform, true);
        }
        rotate(rotationEuler) {
            this.modelTransform = new THREE.Matrix4().makeRotationFromEuler(rotationEuler).multiply(this.modelTransform);
            this.invModelTransform = new THREE.Matrix4().getInverse(this.modelTransform, true);
        }
        scale(scale) {
            this.modelTransform = new THREE.Matrix4().makeScale(scale.x, scale.y, scale.z).multiply(this.modelTransform);
            this.invModelTransform = new THREE.Matrix4().getInverse(this.modelTransform, true);
        }
    }

    // unit shapes in object space (js/8-kd_tree.js:1267-1537)
    class Plane extends Primitive {
        constructor(id) { super(id); this.kind = PLANE; }
        getMinCorner() { return new THREE.Vector3(-10, -10, -10); }   // the handle a plane is selected and grabbed by (js/8-kd_tree.js:1279-1285)
        getMaxCorner() { return new THREE.Vector3(10, 10, 10); }
        scale(scale) { throw new Error("Plane: cannot call abstract method 'scale'"); }
    }
    class Rectangle extends Primitive {
        constructor(id) { super(id); this.kind = RECT; }
        getMinCorner() { return new THREE.Vector3(-0.5, -10, -0.5); }   // js/8-kd_tree.js:1324-1330
        getMaxCorner() { return new THREE.Vector3(0.5, 10, 0.5); }
    }
    class Sphere extends Primitive {
        constructor(id) { super(id); this.kind = SPHERE; }
        getMinCorner() { return new THREE.Vector3(-1, -1, -1); }
        getMaxCorner() { return new THREE.Vector3(1, 1, 1); }
    }
    class Cube extends Primitive {
        constructor(id) { super(id); this.kind = CUBE; }
        getMinCorner() { return new THREE.Vector3(-0.5, -0.5, -0.5); }
        getMaxCorner() { return new THREE.Vector3(0.5, 0.5, 0.5); }
    }
    class Cylinder extends Primitive {
        constructor(id) { super(id); this.kind = CYLINDER; }
        getMinCorner() { return new THREE.Vector3(-1, -0.5, -1); }
        getMaxCorner() { return new THREE.Vector3(1, 0.5, 1); }
    }

    // js/8-kd_tree.js:1977-1990
    class Triangle {
        constructor(v0, v1, v2) {
            this.indices = [v0, v1, v2];
            this.material = null;
        }
        getMaterial() { return this.material; }
        setMaterial(material) { this.material = material; }
    }

    // js/8-kd_tree.js:1996-2117. Vertices are world-space THREE.Vector3 (doubles); every transform takes them
    // back through the single-precision inverse of the old model matrix and forward through the new one.
    class Mesh extends Primitive {
        constructor(id, verts = [], faces = []) {
            super(id);
            this.kind = -1;
            this.verts = verts;
            this.faces = faces;
            this.min = new THREE.Vector3(G.INFINITY, G.INFINITY, G.INFINITY);
            this.max = new THREE.Vector3(-G.INFINITY, -G.INFINITY, -G.INFINITY);
            // the reference's box: x of vertex 0, y of vertex 1, z of vertex 2 of every face
            for (let i = 0; i < faces.length; i++) {
                const idx = faces[i].indices;
                this.min.x = Math.min(this.min.x, verts[idx[0]].x);
                this.max.x = Math.max(this.max.x, verts[idx[0]].x);
                this.min.y = Math.min(this.min.y, verts[idx[1]].y);
                this.max.y = Math.max(this.max.y, verts[idx[1]].y);
                this.min.z = Math.min(this.min.z, verts[idx[2]].z);
                this.max.z = Math.max(this.max.z, verts[idx[2]].z);
            }
            this.pushToGlobal();
        }
        setMaterial(material) {
            for (let t = 0; t < this.faces.length; t++) this.faces[t].setMaterial(material);
        }
        setUniforms(renderer) { super.setUniforms(renderer); }
        getMinCorner() { return this.min; }
        getMaxCorner() { return this.max; }
        _retransform(newModel) {
            this.transformToModel();
            this.modelTransform = newModel;
            this.invModelTransform = new THREE.Matrix4().getInverse(this.modelTransform, true);
            this.transformToWorld();
            this.replaceInGlobal();
        }
        translate(t) { this._retransform(new THREE.Matrix4().makeTranslation(t.x, t.y, t.z).multiply(this.modelTransform)); }
        rotate(e) { this._retransform(new THREE.Matrix4().makeRotationFromEuler(e).multiply(this.modelTransform)); }
        scale(s) { this._retransform(new THREE.Matrix4().makeScale(s.x, s.y, s.z).multiply(this.modelTransform)); }
        transformToWorld() { for (let t = 0; t < this.verts.length; t++) this.verts[t].applyMatrix4(this.modelTransform); }
        transformToModel() { for (let t = 0; t < this.verts.length; t++) this.verts[t].applyMatrix4(this.invModelTransform); }
        // the global arrays hold the same Vector3 / Triangle objects, so a splice of equal length changes nothing
        // unless the mesh was edited; kept because callers may rely on it after editing verts/faces
        replaceInGlobal() {
            // (a splice of 10^5 spread arguments per transform call costs more than the transform itself: skipped when the
            // global arrays already hold exactly these objects in these places, which is the case unless the mesh was edited)
            const holds = (glob, start, oldLength, mine) => {
                if (oldLength !== mine.length) return false;
                for (let i = 0; i < mine.length; i++) if (glob[start + i] !== mine[i]) return false;
                return true;
            };
            if (!holds(G.g_faces, this.startFaceIndex, this.facesLength, this.faces)) G.g_faces.splice(this.startFaceIndex, this.facesLength, ...this.faces);
            if (!holds(G.g_verts, this.startVertIndex, this.vertsLength, this.verts)) G.g_verts.splice(this.startVertIndex, this.vertsLength, ...this.verts);
            this.facesLength = this.faces.length;
            this.vertsLength = this.verts.length;
            G.g_geometryVersion = (G.g_geometryVersion || 0) + 1;
        }
        // NOTE (js/8-kd_tree.js:2108-2116): face indices are appended as they are, not offset by the number of
        // vertices already in g_verts - exactly like the reference, which only ever loads one mesh. Scenes with
        // several meshes pass REBASE_MESH_INDICES = true in the context options (see index.js).
        pushToGlobal() {
            this.startFaceIndex = G.g_faces.length;
            this.startVertIndex = G.g_verts.length;
            this.facesLength = this.faces.length;
            this.vertsLength = this.verts.length;
            if (G.REBASE_MESH_INDICES && this.startVertIndex > 0) {
                for (const f of this.faces) f.indices = f.indices.map((i) => i + this.startVertIndex);
            }
            for (const f of this.faces) G.g_faces.push(f);
            for (const v of this.verts) G.g_verts.push(v);
            G.g_geometryVersion = (G.g_geometryVersion || 0) + 1;
        }
    }

    // js/8-kd_tree.js:2123-2256
    class Material {
        static get PHONG_MATERIAL() { return 0; }
        static get PHONG_BLINN_MATERIAL() { return 1; }
        static get PHONG_CHECKERED_MATERIAL() { return 2; }
        constructor(type, id) {
            if (this.constructor === Material) {
                throw new Error("Material: cannot instantiate abstract class.");
            }
            this.type = type;
            this.id = id;
        }
        getType() { return this.type; }
        getId() { return this.id; }
        setUniforms(renderer) { throw new Error("Material: cannot call abstract method 'setUniforms'"); }
        _put(renderer, array, fields) {
            for (const f of fields) renderer.uniforms[array + "[" + this.id + "]." + f] = this[f];
        }
    }
    const PHONG_FIELDS = ['ka', 'kd', 'ks', 'ke', 'kReflect', 'kRefract', 'ior', 'Co', 'Cs'];
    class Phong extends Material {
        constructor(id, ka, kd, ks, ke, kReflect, kRefract, ior, Co, Cs) {
            super(Material.PHONG_MATERIAL, id);
            Object.assign(this, { ka, kd, ks, ke, kReflect, kRefract, ior, Co, Cs });
        }
        setUniforms(renderer) { this._put(renderer, 'phongMaterials', PHONG_FIELDS); }
    }
    class PhongBlinn extends Material {
        constructor(id, ka, kd, ks, ke, kReflect, kRefract, ior, Co, Cs) {
            super(Material.PHONG_BLINN_MATERIAL, id);
            Object.assign(this, { ka, kd, ks, ke, kReflect, kRefract, ior, Co, Cs });
        }
        setUniforms(renderer) { this._put(renderer, 'phongBlinnMaterials', PHONG_FIELDS); }
    }
    class PhongCheckered extends Material {
        constructor(id, ka, kd, ks, ke, kReflect, kRefract, ior, Co1, Co2, Cs) {
            super(Material.PHONG_CHECKERED_MATERIAL, id);
            Object.assign(this, { ka, kd, ks, ke, kReflect, kRefract, ior, Co1, Co2, Cs });
        }
        setUniforms(renderer) {
            this._put(renderer, 'phongCheckeredMaterials', ['ka', 'kd', 'ks', 'ke', 'kReflect', 'kRefract', 'ior', 'Co1', 'Co2', 'Cs']);
        }
    }

    // js/8-kd_tree.js:2261-2291
    class Light {
        constructor(id) {
            if (this.constructor === Light) {
                throw new Error("Light: cannot instantiate abstract class.");
            }
            this.id = id;
        }
        setUniforms(renderer) { throw new Error("Light: cannot call abstract method 'setUniforms'"); }
    }
    class PointLight extends Light {
        constructor(id, position, color) {
            super(id);
            this.position = position;
            this.color = color;
        }
        setUniforms(renderer) {
            renderer.uniforms["pointLights[" + this.id + "].position"] = this.position;
            renderer.uniforms["pointLights[" + this.id + "].color"] = this.color;
        }
    }

    return { Ray, Primitive, Plane, Rectangle, Sphere, Cube, Cylinder, Triangle, Mesh, Material, Phong, PhongBlinn, PhongCheckered, Light, PointLight };
}

module.exports = { makeSceneAPI, PLANE, RECT, SPHERE, CUBE, CYLINDER };
