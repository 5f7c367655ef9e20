#!/usr/bin/env python3
"""Turns a dump of the reference's generated tracer fragment shader (oracle/gen_golden/ref_capture.js --dump-glsl DIR: the
GLSL ES 3.00 text, or checkpoint 6's ES 1.00 text; the uniform values and the texture payloads of one frame) into a C++ program that EXECUTES that text
on the CPU for a lattice of pixels, and runs it. Build container only; the shader text is read from DIR (outside the
repository) and never copied into it: what is stored under tests/golden/ is the array of colours the program prints.

The C++ side of GLSL is glsl_semantics.h (see its header for what that stand-in does and does not claim). The text
transformations here are purely syntactic:
  * `#version` / `precision` lines dropped, `main` renamed;
  * floating-point literals get an `f` suffix (GLSL literals are float, C++ ones double);
  * `uniform` / `in` / `out` at file scope become plain globals; `inout T x` / `out T x` parameters become `T &x`;
  * `T name[N]` becomes `garray<T, N> name` (out-of-range accesses are dropped, as the oracle assumes);
  * scalar declarations without an initialiser get `= 0` (vector and struct types zero themselves);
  * the three swizzles the shader uses become member calls.
Usage: glsl2cpp.py DUMP_DIR OUT.f32 [--step 4] [--jobs N]   (N processes share the rows; writes OUT.f32: float32 [ceil(H/step), ceil(W/step), 3], rows bottom-up)
       glsl2cpp.py DUMP_DIR OUT.rgba8 --present FRAME.rgba8 LBAR SF
           the reference's present shader (js/8-kd_tree.js:85-143) over an RGBA8 render target of (W*S) x (H*S) bytes*4 with the
           two tone uniforms given as float32 values; writes the canvas, W x H x 4 bytes (float -> byte as the oracle does:
           round(clamp(c, 0, 1) * 255))
"""
import json
import os
import re
import shutil
import struct
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SCALARS = ("float", "int", "uint", "bool")


def f32_literal(v):
    """A float32 value as a C++ expression that reproduces its bits."""
    bits = struct.unpack("<I", struct.pack("<f", float(v)))[0]
    return "fbits(0x%08xu)" % bits


def transform(src):
    src = re.sub(r"^\s*#version.*$", "", src, flags=re.M)
    src = re.sub(r"^\s*precision\s+\w+\s+\w+\s*;\s*$", "", src, flags=re.M)
    src = re.sub(r"//[^\n]*", "", src)
    if "gl_FragColor" in src:
        # GLSL ES 1.00 (js/6-transmission.js): the same stage interface under its older spelling
        src = re.sub(r"^varying\b", "in", src, flags=re.M)
        src = re.sub(r"\bgl_FragColor\b", "fragmentColor", src)
        src = re.sub(r"\btexture2D\s*\(", "texture(", src)
        src = "out vec4 fragmentColor;\n" + src
    # float literals
    src = re.sub(r"(?<![\w.])(\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)(?![\w.])", r"\1f", src)
    structs = re.findall(r"\bstruct\s+(\w+)", src)
    types = ["vec2", "vec3", "vec4", "ivec3", "ivec4", "uvec4", "mat4"] + list(SCALARS) + structs
    tre = "|".join(types)
    # file-scope qualifiers
    uniforms = []

    def uni(m):
        uniforms.append(m.group(3))
        return "%s %s%s;" % (m.group(2), m.group(3), m.group(4) or "")
    src = re.sub(r"^(\s*)uniform\s+(\w+)\s+(\w+)\s*(\[\s*\w+\s*\])?\s*;", uni, src, flags=re.M)
    stage_io = []

    def io(m):
        stage_io.append(m.group(3))
        return "%s %s;" % (m.group(2), m.group(3))
    src = re.sub(r"^(in|out)\s+(\w+)\s+(\w+)\s*;", io, src, flags=re.M)
    # arrays
    src = re.sub(r"\b(%s)\s+(\w+)\s*\[\s*(\w+)\s*\]" % tre, r"garray<\1, \3> \2", src)
    # parameter qualifiers
    src = re.sub(r"\b(?:inout|out)\s+(garray<[^>]+>|\w+)\s+(\w+)", r"\1 &\2", src)
    src = re.sub(r"([(,]\s*)in\s+(garray<[^>]+>|\w+)\s+(\w+)", r"\1\2 \3", src)
    # scalar declarations without any initialiser
    def zero(m):
        names = [n.strip() for n in m.group(2).split(",")]
        return "%s %s;" % (m.group(1), ", ".join(n + " = 0" for n in names))
    src = re.sub(r"\b(%s)\s+((?:\w+\s*,\s*)*\w+)\s*;" % "|".join(SCALARS), zero, src)
    # swizzles
    src = re.sub(r"\.(xyz|yzx|zxy)\b", r".\1()", src)
    src = re.sub(r"\bvoid\s+main\s*\(", "void shader_main(", src)
    # file-scope state that a shader invocation starts afresh with
    depth, pos, reset = 0, 0, []
    for line in src.split("\n"):
        if depth == 0:
            m = re.match(r"^\s*(garray<[^>]+>|%s)\s+(\w+)\s*(=\s*[^;]+)?;" % tre, line)
            if m and m.group(2) not in uniforms and not line.lstrip().startswith("const"):
                reset.append("%s = %s;" % (m.group(2), m.group(3)[1:].strip() if m.group(3) else "%s()" % m.group(1)))
        depth += line.count("{") - line.count("}")
    return src, uniforms, stage_io, reset


def present(dump, out, frame, lbar, sf):
    repo = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
    assert not os.path.abspath(dump).startswith(repo + os.sep), "the dump directory must lie outside the repository"
    with open(os.path.join(dump, "dump.json")) as f:
        meta = json.load(f)
    text = None
    for name in sorted(os.listdir(dump)):
        if name.endswith(".glsl"):
            t = open(os.path.join(dump, name)).read()
            if "sampledColor" in t:
                text = t
    assert text, "no present shader in the dump"
    S = int(re.search(r"const int SUPERSAMPLING = (\d+);", text).group(1))
    body, uniforms, stage_io, reset = transform(text)
    assert set(uniforms) == {"tex", "sf", "Lbar"} and "texCoord" in stage_io, (uniforms, stage_io)
    W, H = meta["width"], meta["height"]
    cpp = r'''// generated by glsl2cpp.py - scratch file, never committed (it holds the reference's shader text)
#include <cstdio>
#include <cstdlib>
#undef INFINITY
namespace shader {
#include "%(hdr)s"
static inline float fbits(unsigned int b) { float f; __builtin_memcpy(&f, &b, 4); return f; }
%(body)s
}  // namespace shader
static unsigned char quant8(float c) {
    if (!(c == c)) return 0;
    float f = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
    return (unsigned char)(int)(f * 255.0f + 0.5f);
}
int main(int argc, char **argv) {
    using namespace shader;
    const int W = %(W)d, H = %(H)d, S = %(S)d;
    size_t n = (size_t)W * S * H * S * 4;
    unsigned char *frame = (unsigned char *)malloc(n);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(frame, 1, n, f) != n) { fprintf(stderr, "cannot read the frame\n"); return 1; }
    fclose(f);
    g_units[0] = TexImage{frame, W * S, H * S, 4, true};
    FILE *out = fopen(argv[2], "wb");
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            tex = 0;
            Lbar = fbits(%(lbar)s);
            sf = fbits(%(sf)s);
            texCoord = vec2(((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H);
            fragmentColor = vec4();
            shader_main();
            unsigned char px[4] = {quant8(fragmentColor.x), quant8(fragmentColor.y), quant8(fragmentColor.z), 255};
            fwrite(px, 1, 4, out);
        }
    fclose(out);
    return 0;
}
''' % {"hdr": os.path.join(HERE, "glsl_semantics.h"), "body": body, "W": W, "H": H, "S": S,
       "lbar": "0x%08xu" % struct.unpack("<I", struct.pack("<f", float(lbar)))[0],
       "sf": "0x%08xu" % struct.unpack("<I", struct.pack("<f", float(sf)))[0]}
    src_path, exe = os.path.join(dump, "present_exec.cpp"), os.path.join(dump, "present_exec")
    with open(src_path, "w") as f:
        f.write(cpp)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused", "-o", exe, src_path])
    subprocess.check_call([exe, frame, out])
    print("wrote", out, "(%d x %d canvas, supersampling %d)" % (W, H, S))


def main():
    dump, out = sys.argv[1], sys.argv[2]
    if "--present" in sys.argv:
        i = sys.argv.index("--present")
        return present(dump, out, sys.argv[i + 1], float(sys.argv[i + 2]), float(sys.argv[i + 3]))
    step = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 4
    repo = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
    assert not os.path.abspath(dump).startswith(repo + os.sep), "the dump directory must lie outside the repository"
    with open(os.path.join(dump, "dump.json")) as f:
        meta = json.load(f)
    frag = None
    for name in sorted(os.listdir(dump)):
        if name.endswith(".glsl"):
            text = open(os.path.join(dump, name)).read()
            if "castRayRecursive" in text:
                frag = text
    assert frag, "no tracer fragment shader in the dump"
    body, uniforms, stage_io, reset = transform(frag)
    assert "fragmentColor" in stage_io and "primaryRayDir" in stage_io, stage_io

    # uniform values: every recorded gl.uniform* call whose root name the shader declares
    assign, corner = [], {}
    for u in meta["uniforms"]:
        root = re.match(r"\w+", u["name"]).group(0)
        v = u["value"]
        if root in ("ray00", "ray01", "ray10", "ray11"):
            corner[root] = v
        if root not in uniforms:
            continue
        if u["fn"] == "uniform1i":
            assign.append("%s = %d;" % (u["name"], int(v[0])))
        elif u["fn"] == "uniform1f":
            assign.append("%s = %s;" % (u["name"], f32_literal(v[0])))
        elif u["fn"] == "uniform3fv":
            assign.append("%s = vec3(%s);" % (u["name"], ", ".join(f32_literal(x) for x in v)))
        elif u["fn"] == "uniformMatrix4fv":
            assign += ["%s.m[%d] = %s;" % (u["name"], i, f32_literal(x)) for i, x in enumerate(v)]
        else:
            raise SystemExit("unhandled uniform call " + u["fn"])
    assert len(corner) == 4
    # textures: the last texSubImage2D payload of every unit. `gl.TEXTURE1 + i` on the recording context, whose constants
    # are their own names, spells units 1 and 2 as "TEXTURE10" and "TEXTURE11"
    chans = {"RGB": 3, "RGB_INTEGER": 3, "RED_INTEGER": 1, "RGBA_INTEGER": 4, "RGBA": 4}
    tex = {}
    for t in meta["textures"]:
        if t["call"] != "texSubImage2D":
            continue
        digits = t["unit"][len("TEXTURE"):]
        unit = int(digits[0]) + int(digits[1]) if len(digits) == 2 else int(digits)
        tex[unit] = (os.path.join(os.path.abspath(dump), t["file"]), t["args"][4], t["args"][5], chans[t["args"][6]])
    loads = ["load_unit(%d, \"%s\", %d, %d, %d);" % (u, p, w, h, c) for u, (p, w, h, c) in sorted(tex.items())]

    W, H = int(meta["traced_width"]), int(meta["traced_height"])
    cpp = r'''// generated by glsl2cpp.py - scratch file, never committed (it holds the reference's shader text)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#undef INFINITY
namespace shader {
#include "%(hdr)s"
static inline float fbits(unsigned int b) { float f; __builtin_memcpy(&f, &b, 4); return f; }
%(body)s
static void set_uniforms() {
%(assign)s
}
static void reset_globals() {
%(reset)s
}
static void load_unit(int unit, const char *path, int w, int h, int c) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    size_t n = (size_t)w * h * c * 4;
    void *p = malloc(n);
    if (fread(p, 1, n, f) != n) { fprintf(stderr, "short read: %%s\n", path); exit(1); }
    fclose(f);
    g_units[unit] = TexImage{p, w, h, c, false};
}
}  // namespace shader

int main(int argc, char **argv) {
    using namespace shader;
    %(loads)s
    const int W = %(W)d, H = %(H)d, step = %(step)d;
    const vec3 r00(%(r00)s), r01(%(r01)s), r10(%(r10)s), r11(%(r11)s);
    FILE *out = fopen(argv[1], "wb");
    if (!out) { perror(argv[1]); return 1; }
    // optional row range (lattice rows y0 <= y < y1): several processes share a large frame
    const int y0 = argc > 3 ? atoi(argv[2]) : 0, y1 = argc > 3 ? atoi(argv[3]) : H;
    for (int y = y0; y < y1 && y < H; y += step)
        for (int x = 0; x < W; x += step) {
            reset_globals();
            set_uniforms();
            // the vertex stage and the rasteriser: the corner rays mixed at the pixel centre (js/8-kd_tree.js:177-190)
            float fx = ((float)x + 0.5f) / (float)W, fy = ((float)y + 0.5f) / (float)H;
            primaryRayDir = mix(mix(r00, r01, fy), mix(r10, r11, fy), fx);
            shader_main();
            float c[3] = {fragmentColor.x, fragmentColor.y, fragmentColor.z};
            fwrite(c, 4, 3, out);
        }
    fclose(out);
    return 0;
}
''' % {"hdr": os.path.join(HERE, "glsl_semantics.h"), "body": body, "assign": "\n".join("    " + a for a in assign),
       "reset": "\n".join("    " + r for r in reset), "loads": "\n    ".join(loads), "W": W, "H": H, "step": step,
       "r00": ", ".join(f32_literal(x) for x in corner["ray00"]), "r01": ", ".join(f32_literal(x) for x in corner["ray01"]),
       "r10": ", ".join(f32_literal(x) for x in corner["ray10"]), "r11": ", ".join(f32_literal(x) for x in corner["ray11"])}
    cpp = cpp.replace("r00(fbits", "r00(shader::fbits")   # (fbits lives in the namespace; `using` covers it, kept explicit for the first)
    src_path = os.path.join(dump, "tracer_exec.cpp")
    exe = os.path.join(dump, "tracer_exec")
    with open(src_path, "w") as f:
        f.write(cpp)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused", "-o", exe, src_path])
    jobs = int(sys.argv[sys.argv.index("--jobs") + 1]) if "--jobs" in sys.argv else 1
    if jobs <= 1:
        subprocess.check_call([exe, out])
    else:
        # row slices that start on lattice rows, one process each, outputs joined in order
        n_rows = (H + step - 1) // step
        cuts = [((n_rows * k) // jobs) * step for k in range(jobs)] + [H]
        procs = [subprocess.Popen([exe, "%s.part%d" % (out, k), str(cuts[k]), str(cuts[k + 1])]) for k in range(jobs)]
        for pr in procs:
            if pr.wait() != 0:
                raise SystemExit("a slice of the frame failed")
        with open(out, "wb") as f:
            for k in range(jobs):
                with open("%s.part%d" % (out, k), "rb") as g:
                    shutil.copyfileobj(g, f)
                os.remove("%s.part%d" % (out, k))
    print("wrote", out, "(%d x %d samples of the %d x %d frame)" % ((W + step - 1) // step, (H + step - 1) // step, W, H))


if __name__ == "__main__":
    main()
