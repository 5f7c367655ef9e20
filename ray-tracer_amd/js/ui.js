'use strict';
// Selecting and dragging objects with the mouse: the arithmetic of the reference's UI.mouseDown / mouseMove / mouseUp
// (js/8-kd_tree.js:2718-2792) on top of RayTracer.pick. What differs from the reference:
//   - the first click finds the object under the pixel in the primary hit-id buffer of the last traced frame
//     (rt_pick) instead of intersecting the pixel's ray with every object on the CPU (Primitive.intersect,
//     js/8-kd_tree.js:2746-2752). For a Mesh the reference tests the mesh's bounding box, so a click on an empty
//     corner of that box selects the mesh there and whatever is visible at that pixel here.
//   - no selection-box overlay is drawn and there are no camera controls to disable.
// The press on the selected object's box that starts a drag (Cube.intersect, js/8-kd_tree.js:1439-1454), the depth at
// which the object follows the mouse and every translation are computed exactly as the reference does, in doubles
// with single-precision matrices, so a dragged scene ends up with bit-identical transforms and vertices.

function makeUI(G, THREE, api) {
    // js/8-kd_tree.js:1439-1454: distance along `ray` to the box [minCorner, maxCorner] of an object, in its own space
    function boxDistance(ray, minCorner, maxCorner, invModelTransform) {
        const localRayOrigin = ray.origin.clone().applyMatrix4(invModelTransform);
        const localRayDir = ray.dir.clone().transformDirection(invModelTransform);
        const absLocalRayDir = new THREE.Vector3(Math.abs(localRayDir.x), Math.abs(localRayDir.y), Math.abs(localRayDir.z));
        const t0 = minCorner.clone().divide(absLocalRayDir).sub(localRayOrigin.clone().divide(localRayDir));
        const t1 = maxCorner.clone().divide(absLocalRayDir).sub(localRayOrigin.clone().divide(localRayDir));
        const tMin = Math.max(t0.x, t0.y, t0.z), tMax = Math.min(t1.x, t1.y, t1.z);
        if (tMin > tMax || tMax < 0.0) return Number.MAX_VALUE;
        return tMin;
    }

    class UI {
        // tracer: a RayTracer whose last update() rendered the scene that is being edited
        constructor(tracer) {
            this.tracer = tracer;
            this.scene = tracer.scene;
            this.selectedObject = null;
            this.moving = false;
        }
        setScene(scene) { this.scene = scene; this.selectedObject = null; this.tracer.setScene(scene); }
        select(object) { this.selectedObject = object; }   // what a click on `object` does (for callers without a traced frame)

        // js/8-kd_tree.js:2718-2761; x, y in canvas pixels, y from the top. Returns true when the press landed on something.
        mouseDown(x, y) {
            const camera = G.camera;
            const ray = new api.Ray(camera.position.clone(), G.getPrimaryRay((x / G.WIDTH) * 2 - 1, 1 - (y / G.HEIGHT) * 2));
            if (this.selectedObject != null) {   // a press on the selection box grabs the object
                const so = this.selectedObject;
                const t = boxDistance(ray, so.getMinCorner(), so.getMaxCorner(), so.invModelTransform);
                if (t < Number.MAX_VALUE) {
                    const position = new THREE.Vector3().setFromMatrixPosition(so.modelTransform);
                    this.oldPosition = position.clone();
                    position.applyMatrix4(new THREE.Matrix4().getInverse(camera.matrixWorld));
                    this.zPos = position.z;
                    this.moving = true;
                    return true;
                }
            }
            this.selectedObject = this.tracer.pick(x, y);
            return this.selectedObject != null;
        }

        // the point under the mouse at the grabbed object's depth (js/8-kd_tree.js:2765-2768)
        _follow(x, y) {
            const camera = G.camera;
            const pos = new THREE.Vector3((x / G.WIDTH) * 2 - 1, 1 - (y / G.HEIGHT) * 2, 0.5).unproject(camera);
            const dir = pos.sub(camera.position).normalize().transformDirection(new THREE.Matrix4().getInverse(camera.matrixWorld));
            const distance = this.zPos / dir.z;
            const newPosition = dir.multiplyScalar(distance).applyMatrix4(camera.matrixWorld);
            this.selectedObject.translate(newPosition.clone().sub(this.oldPosition));
            this.oldPosition = newPosition;
        }
        mouseMove(x, y) { if (this.moving) this._follow(x, y); }      // js/8-kd_tree.js:2763-2775
        mouseUp(x, y) {                                               // js/8-kd_tree.js:2777-2792
            if (this.moving) {
                this._follow(x, y);
                this.moving = false;
            }
        }
    }
    return { UI, boxDistance };
}

module.exports = { makeUI };
