'use strict';
// The interactive front end of the reference on top of class RayTracer: class UI with update(time) / render(),
// mouseDown / mouseMove / mouseUp (js/8-kd_tree.js:2672-2803), the selection box Renderer.render() draws round the
// selected object (:2646-2664, line shaders :149-171), and the event plumbing of the page (:3038-3088). There is no
// browser here, so:
//   - `canvas`, `document` and `window` are small event surfaces (EventSurface below). three.js's own
//     examples/js/controls/TrackballControls.js - the camera controls the reference's UI creates (:2680-2688) - runs
//     unchanged against them and against three_lite.js: evaluate that file in the context and the UI picks
//     THREE.TrackballControls up (attachControls). A front end (examples/serve.js streams the canvas to a browser) hands
//     the browser's mouse events to UI.handleEvent, which delivers them in the page's order: the page's own canvas
//     handlers first, then the listeners the controls registered.
//   - the first click finds the object under the pixel in the primary hit-id buffer of the last traced frame
//     (rt_pick) instead of intersecting the pixel's ray with every object on the CPU (Primitive.intersect, :2746-2752).
//     For a Mesh the reference tests the mesh's bounding box, so a click on an empty corner of that box selects the
//     mesh there and whatever is visible at that pixel here.
//   - the selection box is drawn into the canvas pixels by a line rasteriser (drawSelectionBox): the clip-space
//     corners are the reference's (same uniforms, same vertex-shader arithmetic in single precision), which pixels a GL
//     implementation lights along an edge is its own affair.
// The press on the selected object's box that starts a drag (Cube.intersect, :1439-1454), the depth at which the
// object follows the mouse and every translation are computed exactly as the reference does, in doubles with
// single-precision matrices, so a dragged scene ends up with bit-identical transforms and vertices.

// the 12 edges of the unit cube as pairs of corner numbers; corner c = (c & 1, c >> 1 & 1, c >> 2) (js/8-kd_tree.js:2599-2613)
const BOX_EDGES = [0, 1, 1, 3, 3, 2, 2, 0, 4, 5, 5, 7, 7, 6, 6, 4, 0, 4, 1, 5, 2, 6, 3, 7];

// listeners by event type, as an element or a document keeps them
class EventSurface {
    constructor() { this._listeners = {}; }
    addEventListener(type, fn) {
        const l = this._listeners[type] = this._listeners[type] || [];
        if (l.indexOf(fn) < 0) l.push(fn);
    }
    removeEventListener(type, fn) {
        const l = this._listeners[type] || [], i = l.indexOf(fn);
        if (i >= 0) l.splice(i, 1);
    }
    dispatch(type, event) { for (const fn of (this._listeners[type] || []).slice()) fn(event); }
}

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

    // what the page script finds in a browser: the canvas element, `document`, and `window` (the context itself)
    if (!G.canvas) {
        const canvas = new EventSurface();
        canvas.getBoundingClientRect = () => ({ left: 0, top: 0, width: +G.WIDTH, height: +G.HEIGHT });
        canvas.style = {};
        const document = new EventSurface();
        document.documentElement = { clientLeft: 0, clientTop: 0, scrollLeft: 0, scrollTop: 0 };
        document.body = { scrollLeft: 0, scrollTop: 0 };
        document.getElementById = () => canvas;
        canvas.ownerDocument = document;
        const win = new EventSurface();
        G.canvas = canvas;
        G.document = document;
        G.addEventListener = win.addEventListener.bind(win);
        G.removeEventListener = win.removeEventListener.bind(win);
        G.windowEvents = win;
        G.pageXOffset = 0;
        G.pageYOffset = 0;
        G.innerWidth = +G.WIDTH;
        G.innerHeight = +G.HEIGHT;
        G.window = G;
    }

    class UI {
        // tracer: a RayTracer (the reference's UI owns a Renderer that owns the RayTracer, js/8-kd_tree.js:2676, :2632)
        constructor(tracer) {
            this.tracer = tracer;
            this.scene = tracer.scene;
            this.selectedObject = null;
            this.moving = false;
            this.controls = null;
            this.viewProjection = new THREE.Matrix4();
            this._pageMouseDown = false;   // `mouseDown` of the page script (js/8-kd_tree.js:3054)
            this.attachControls();
        }

        // js/8-kd_tree.js:2679-2688; ROTATE / ZOOM / PAN are the constants of :17-19. Without THREE.TrackballControls in the
        // context (the file was not evaluated) the camera stays where the caller puts it.
        attachControls() {
            if (typeof THREE.TrackballControls !== 'function') return null;
            this.controls = new THREE.TrackballControls(G.camera, G.canvas);
            this.controls.rotateSpeed = G.ROTATE === undefined ? 15.0 : G.ROTATE;
            this.controls.zoomSpeed = G.ZOOM === undefined ? 10.0 : G.ZOOM;
            this.controls.panSpeed = G.PAN === undefined ? 6.0 : G.PAN;
            this.controls.noZoom = false;
            this.controls.noPan = false;
            this.controls.staticMoving = false;
            this.controls.dynamicDampingFactor = 0.3;
            this.resetControls();
            return this.controls;
        }
        resetControls() {   // js/8-kd_tree.js:2798-2802
            this.controls.reset();
            this.controls.target.set(0, 0, -270);
            G.camera.updateMatrixWorld(true);
        }

        setScene(scene) { this.scene = scene; this.selectedObject = null; this.tracer.setScene(scene); }
        select(object) { this.selectedObject = object; }   // what a click on `object` does (for callers without a traced frame)

        // js/8-kd_tree.js:2699-2709 and Renderer.update (:2641-2644): light animation, camera controls, camera matrices,
        // view-projection for the selection box, then the frame
        update(time) {
            G.animate(this.scene, time);
            this.prepareCamera();
            this.tracer.update(time);
        }
        prepareCamera() {
            if (this.controls) this.controls.update();
            G.camera.updateMatrixWorld(true);
            G.camera.updateProjectionMatrix();
            this.viewProjection = new THREE.Matrix4().multiplyMatrices(G.camera.projectionMatrix, new THREE.Matrix4().getInverse(G.camera.matrixWorld));
            return this.viewProjection;
        }

        // js/8-kd_tree.js:2794-2796 and Renderer.render (:2646-2664): the present pass, then the selection box on top
        render() {
            const pixels = this.tracer.render();
            if (this.selectedObject != null) this.drawSelectionBox(pixels, this.tracer.width, this.tracer.height);
            return pixels;
        }

        // the uniforms of the line program (js/8-kd_tree.js:2655-2661), narrowed to single precision as setUniforms does
        selectionBox() {
            if (this.selectedObject == null) return null;
            const so = this.selectedObject;
            const lo = so.getMinCorner(), hi = so.getMaxCorner();
            return {
                cubeMin: new Float32Array([lo.x, lo.y, lo.z]),
                cubeMax: new Float32Array([hi.x, hi.y, hi.z]),
                modelViewProjection: new Float32Array(new THREE.Matrix4().multiplyMatrices(this.viewProjection, so.modelTransform).elements),
            };
        }

        // The 12 edges after the line vertex shader (gl_Position = modelViewProjection * vec4(mix(cubeMin, cubeMax, vertex), 1),
        // js/8-kd_tree.js:149-158, in single precision), clipped to w > 0 and to the canvas, in pixel coordinates with y
        // counted from the top. Returns [[x0, y0, x1, y1], ...].
        selectionBoxSegments(width, height) {
            const box = this.selectionBox();
            if (!box) return [];
            const f = Math.fround, m = box.modelViewProjection, lo = box.cubeMin, hi = box.cubeMax;
            const clip = [];
            for (let c = 0; c < 8; c++) {
                const v = [c & 1, (c >> 1) & 1, c >> 2];
                // mix(a, b, t) = a * (1 - t) + b * t with t = 0 or 1
                const p = v.map((t, k) => f(f(lo[k] * f(1 - t)) + f(hi[k] * t)));
                const row = (r) => f(f(f(f(m[r] * p[0]) + f(m[4 + r] * p[1])) + f(m[8 + r] * p[2])) + m[12 + r]);
                clip.push([row(0), row(1), row(2), row(3)]);
            }
            const out = [];
            for (let e = 0; e < BOX_EDGES.length; e += 2) {
                let a = clip[BOX_EDGES[e]], b = clip[BOX_EDGES[e + 1]];
                // the six clip planes w +- x, w +- y, w +- z >= 0 (Liang-Barsky on the homogeneous segment)
                let t0 = 0, t1 = 1, visible = true;
                for (let k = 0; k < 3 && visible; k++) {
                    for (const s of [1, -1]) {
                        const da = a[3] + s * a[k], db = b[3] + s * b[k];
                        if (da < 0 && db < 0) { visible = false; break; }
                        if (da < 0) t0 = Math.max(t0, da / (da - db));
                        else if (db < 0) t1 = Math.min(t1, da / (da - db));
                    }
                }
                if (!visible || t0 > t1) continue;
                const at = (t) => a.map((x, i) => x + (b[i] - x) * t);
                const pa = at(t0), pb = at(t1);
                const px = (p) => [(p[0] / p[3] * 0.5 + 0.5) * width, (1 - (p[1] / p[3] * 0.5 + 0.5)) * height];
                out.push([...px(pa), ...px(pb)]);
            }
            return out;
        }

        // white one-pixel lines (the line fragment shader writes vec4(1.0), js/8-kd_tree.js:164-171) into an RGBA8 canvas.
        // RayTracer.render() returns rows bottom-up, as readPixels does; pass bottomUp = false for a top-down image
        drawSelectionBox(pixels, width, height, bottomUp) {
            const flip = bottomUp === undefined ? true : bottomUp;
            const put = (x, y) => {
                if (x < 0 || y < 0 || x >= width || y >= height) return;
                const o = 4 * ((flip ? height - 1 - y : y) * width + x);
                pixels[o] = pixels[o + 1] = pixels[o + 2] = pixels[o + 3] = 255;
            };
            for (const [x0, y0, x1, y1] of this.selectionBoxSegments(width, height)) {
                const dx = x1 - x0, dy = y1 - y0;
                const steps = Math.max(1, Math.ceil(Math.max(Math.abs(dx), Math.abs(dy))));
                for (let i = 0; i <= steps; i++) put(Math.floor(x0 + dx * i / steps), Math.floor(y0 + dy * i / steps));
            }
            return pixels;
        }

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
                    if (this.controls) this.controls.enabled = false;
                    return true;
                }
            }
            this.selectedObject = this.tracer.pick(x, y);
            if (this.controls) this.controls.enabled = this.selectedObject == null;
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
            if (this.controls) this.controls.enabled = true;
        }

        // One browser event, delivered as the page delivers it: first the page's own canvas.onmousedown / onmousemove /
        // onmouseup (js/8-kd_tree.js:3056-3088), then the listeners registered on the canvas (the camera controls) and,
        // for moves and releases, those on the document (the controls listen there while a button is down).
        // type: 'mousedown' | 'mousemove' | 'mouseup' | 'wheel' | 'contextmenu'; event: { x, y, button, deltaY, deltaMode }
        // with x, y in canvas pixels from the top-left corner.
        handleEvent(type, event) {
            const ev = Object.assign({ button: 0, deltaY: 0, deltaMode: 0 }, event, {
                type, pageX: event.x, pageY: event.y, clientX: event.x, clientY: event.y, preventDefault() {}, stopPropagation() {},
            });
            const x = event.x, y = event.y;
            if (type === 'mousedown') {
                if (x >= 0 && x < G.WIDTH && y >= 0 && y < G.HEIGHT) this._pageMouseDown = !this.mouseDown(x, y);
            } else if (type === 'mousemove') {
                if (!this._pageMouseDown) this.mouseMove(x, y);
            } else if (type === 'mouseup') {
                this._pageMouseDown = false;
                this.mouseUp(x, y);
            }
            G.canvas.dispatch(type, ev);
            if (type === 'mousemove' || type === 'mouseup') G.document.dispatch(type, ev);
        }
    }
    return { UI, boxDistance, EventSurface };
}

module.exports = { makeUI, EventSurface, BOX_EDGES };
