'use strict';
// The reference is a web page: a canvas that shows the traced frame, mouse picking and dragging, camera controls. This
// is that page served from the machine with the GPU: an HTTP server that traces on the GPU, streams the presented canvas
// (with the selection box) to a browser as PNG frames and takes the browser's mouse events back.
//   node serve.js [--port 8080] [--host 127.0.0.1] [--width 640 --height 480] [--depth 3] [--model models/...ply]
//                 [--kd-builder device] [--controls path/to/three.js/examples/js/controls/TrackballControls.js]
//                 [--self-test: talk to the server over HTTP from this process, print a JSON summary, exit]
//   GET  /            the page: an <img> fed by /stream, listeners that POST mouse events to /event
//   GET  /frame.png   one frame, traced now
//   GET  /stream      multipart/x-mixed-replace, one PNG per tick (about 60 a second at most)
//   POST /event       {"type": "mousedown"|"mousemove"|"mouseup"|"wheel", "x":, "y":, "button":, "deltaY":}
// Camera controls need three.js's own TrackballControls.js (--controls): it runs unchanged on three_lite.js.
const fs = require('fs');
const http = require('http');
const path = require('path');
const crypto = require('crypto');
const rt = require('..');
const { encodePNG, decodePNG } = require('../png.js');

const arg = (name, dflt) => { const i = process.argv.indexOf('--' + name); return i > 0 ? process.argv[i + 1] : dflt; };
const flag = (name) => process.argv.indexOf('--' + name) > 0;
const opts = {
    WIDTH: parseInt(arg('width', '640'), 10), HEIGHT: parseInt(arg('height', '480'), 10),
    MAX_RECURSION_DEPTH: parseInt(arg('depth', '3'), 10), MODEL_FILE_NAME: arg('model', 'models/bun_zipper_res4.ply'),
    DEVICE: parseInt(arg('device', '0'), 10), KD_BUILDER: arg('kd-builder', 'device'), KD_REBUILD_INTERVAL: 0,
};
const mm = arg('mesh-material', null);
if (mm) opts.MESH_MATERIAL = mm.split(',').map(parseFloat);
const G = rt.createContext(opts);
const sceneFile = arg('scene', path.join(__dirname, 'default_scene.js'));
rt.run(G, fs.readFileSync(sceneFile, 'utf8'), sceneFile);
const controlsFile = arg('controls', null);
if (controlsFile) rt.run(G, fs.readFileSync(controlsFile, 'utf8'), controlsFile);

const PAGE = `<!doctype html><html><head><meta charset="utf-8"><title>ray tracer</title></head>
<body style="margin:0;background:#222"><img id="v" src="/stream" width="${opts.WIDTH}" height="${opts.HEIGHT}" draggable="false" style="display:block">
<script>
var v = document.getElementById('v');
function send(type, e) {
  var r = v.getBoundingClientRect();
  var x = new XMLHttpRequest();
  x.open('POST', '/event');
  x.send(JSON.stringify({type: type, x: e.clientX - r.left, y: e.clientY - r.top, button: e.button, deltaY: e.deltaY || 0, deltaMode: e.deltaMode || 0}));
  e.preventDefault();
}
v.addEventListener('mousedown', function (e) { send('mousedown', e); });
document.addEventListener('mousemove', function (e) { if (e.buttons) send('mousemove', e); });
document.addEventListener('mouseup', function (e) { send('mouseup', e); });
v.addEventListener('wheel', function (e) { send('wheel', e); });
v.addEventListener('contextmenu', function (e) { e.preventDefault(); });
</script></body></html>`;

(async () => {
    const data = await G.loadPLY(G.MODEL_FILE_NAME);
    const scene = G.generateScene({ [G.MODEL_FILE_NAME]: data });
    const tracer = new G.RayTracer();
    tracer.setScene(scene);
    const ui = new G.UI(tracer);
    const start = Date.now();
    let fixedTime = null;   // the self-test renders at stated times
    const frame = () => {
        ui.update(fixedTime !== null ? fixedTime : (Date.now() - start) * 0.001);   // js/8-kd_tree.js:2829-2849
        return ui.render();
    };
    const png = () => encodePNG(frame(), G.WIDTH, G.HEIGHT, true);

    const server = http.createServer((req, res) => {
        if (req.method === 'GET' && req.url === '/') {
            res.writeHead(200, { 'Content-Type': 'text/html' });
            res.end(PAGE);
        } else if (req.method === 'GET' && req.url.startsWith('/frame.png')) {
            const body = png();
            res.writeHead(200, { 'Content-Type': 'image/png', 'Content-Length': body.length, 'Cache-Control': 'no-store' });
            res.end(body);
        } else if (req.method === 'GET' && req.url === '/stream') {
            res.writeHead(200, { 'Content-Type': 'multipart/x-mixed-replace; boundary=frame', 'Cache-Control': 'no-store' });
            let open = true;
            req.on('close', () => { open = false; });
            const next = () => {
                if (!open) return;
                const body = png();
                res.write('--frame\r\nContent-Type: image/png\r\nContent-Length: ' + body.length + '\r\n\r\n');
                res.write(body);
                res.write('\r\n', () => setTimeout(next, 1000 / 60));
            };
            next();
        } else if (req.method === 'POST' && req.url === '/event') {
            const parts = [];
            let size = 0;
            req.on('data', (d) => {
                size += d.length;
                if (size > 4096) { res.writeHead(413); res.end(); req.destroy(); return; }   // an event is a few dozen bytes
                parts.push(d);
            });
            req.on('end', () => {
                if (size > 4096) return;
                try {
                    const ev = JSON.parse(Buffer.concat(parts).toString('utf8'));
                    if (!/^(mousedown|mousemove|mouseup|wheel)$/.test(ev.type)) throw new Error('unknown event type');
                    ui.handleEvent(ev.type, { x: +ev.x, y: +ev.y, button: ev.button | 0, deltaY: +ev.deltaY || 0, deltaMode: ev.deltaMode | 0 });
                    res.writeHead(200, { 'Content-Type': 'application/json' });
                    res.end(JSON.stringify({ selected: ui.selectedObject ? ui.selectedObject.constructor.name : null, moving: ui.moving }));
                } catch (e) {
                    res.writeHead(400, { 'Content-Type': 'text/plain' });
                    res.end(String(e.message || e));
                }
            });
        } else {
            res.writeHead(404);
            res.end();
        }
    });

    const host = arg('host', '127.0.0.1');
    await new Promise((resolve) => server.listen(flag('self-test') ? 0 : parseInt(arg('port', '8080'), 10), host, resolve));
    const port = server.address().port;
    if (!flag('self-test')) {
        console.log('serving http://' + host + ':' + port + '/');
        return;
    }

    // ---- self-test: a client in this process does over HTTP what a browser would ----------------------------------
    const request = (method, url, body) => new Promise((resolve, reject) => {
        const r = http.request({ host, port, path: url, method }, (res) => {
            const parts = [];
            res.on('data', (d) => parts.push(d));
            res.on('end', () => resolve({ status: res.statusCode, type: res.headers['content-type'], body: Buffer.concat(parts) }));
        });
        r.on('error', reject);
        if (body) r.write(body);
        r.end();
    });
    const sha = (u8) => crypto.createHash('sha256').update(Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength)).digest('hex');
    const topDown = (canvas) => {   // what the PNG holds: rows top-down
        const W = G.WIDTH, H = G.HEIGHT, out = new Uint8Array(W * H * 4);
        for (let y = 0; y < H; y++) out.set(canvas.subarray((H - 1 - y) * W * 4, (H - y) * W * 4), y * W * 4);
        return out;
    };
    const out = {};
    fixedTime = 0;
    const page = await request('GET', '/');
    out.page_ok = page.status === 200 && /<img id="v" src="\/stream"/.test(page.body.toString('utf8'));
    const f0 = decodePNG((await request('GET', '/frame.png')).body);
    out.frame_size = [f0.width, f0.height];
    out.plain_frame_equals_direct_render = sha(f0.pixels) === sha(topDown(frame()));
    // a click on the glass sphere in the middle of the canvas selects it: the next frame carries its selection box
    // (not the exact centre: that pixel's ray runs along -z, and the reference's box test, js/8-kd_tree.js:1439-1454,
    // divides by the zero direction components and never grabs there)
    const clickAt = [G.WIDTH / 2 + 7, G.HEIGHT / 2 - 5];
    const r1 = JSON.parse((await request('POST', '/event', JSON.stringify({ type: 'mousedown', x: clickAt[0], y: clickAt[1], button: 0 }))).body.toString('utf8'));
    await request('POST', '/event', JSON.stringify({ type: 'mouseup', x: clickAt[0], y: clickAt[1], button: 0 }));
    out.selected = r1.selected;
    const f1 = decodePNG((await request('GET', '/frame.png')).body);
    let white = 0, changed = 0;
    for (let i = 0; i < f1.pixels.length; i += 4) {
        const w = f1.pixels[i] === 255 && f1.pixels[i + 1] === 255 && f1.pixels[i + 2] === 255;
        const same = f1.pixels[i] === f0.pixels[i] && f1.pixels[i + 1] === f0.pixels[i + 1] && f1.pixels[i + 2] === f0.pixels[i + 2];
        if (!same) { changed++; if (w) white++; }
    }
    out.overlay_pixels = white;
    out.pixels_changed_by_overlay_only = changed === white;
    out.selection_segments = ui.selectionBoxSegments(G.WIDTH, G.HEIGHT).length;
    // grab it (a second press on its box), drag it 40 pixels to the right, release: the next frame differs and the box moved along
    const before = Array.from(ui.selectedObject.modelTransform.elements);
    const r2 = JSON.parse((await request('POST', '/event', JSON.stringify({ type: 'mousedown', x: clickAt[0], y: clickAt[1], button: 0 }))).body.toString('utf8'));
    out.grabbed = r2.moving;
    await request('POST', '/event', JSON.stringify({ type: 'mousemove', x: clickAt[0] + 40, y: clickAt[1], button: 0 }));
    await request('POST', '/event', JSON.stringify({ type: 'mouseup', x: clickAt[0] + 40, y: clickAt[1], button: 0 }));
    const after = Array.from(ui.selectedObject.modelTransform.elements);
    out.moved_x = after[12] - before[12];
    const f2 = decodePNG((await request('GET', '/frame.png')).body);
    out.frame_after_drag_differs = sha(f2.pixels) !== sha(f1.pixels);
    out.frame_after_drag_equals_direct_render = sha(f2.pixels) === sha(topDown(frame()));
    // the first part of the stream is a PNG of the same frame
    out.stream_first_part_png = await new Promise((resolve, reject) => {
        const r = http.request({ host, port, path: '/stream', method: 'GET' }, (res) => {
            let buf = Buffer.alloc(0);
            res.on('data', (d) => {
                buf = Buffer.concat([buf, d]);
                const h = buf.indexOf('\r\n\r\n');
                if (h < 0) return;
                const m = /Content-Length: (\d+)/.exec(buf.toString('latin1', 0, h));
                if (m && buf.length >= h + 4 + (+m[1])) {
                    const img = decodePNG(buf.slice(h + 4, h + 4 + (+m[1])));
                    res.destroy();
                    resolve(sha(img.pixels) === sha(f2.pixels));
                }
            });
        });
        r.on('error', reject);
        r.end();
    });
    out.bad_event_status = (await request('POST', '/event', JSON.stringify({ type: 'keydown' }))).status;
    console.log(JSON.stringify(out));
    server.close();
    tracer.dispose();
    process.exit(0);
})().catch((e) => { console.error(e.stack || e); process.exit(1); });
