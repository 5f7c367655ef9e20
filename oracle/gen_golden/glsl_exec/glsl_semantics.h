// Stand-in for the parts of GLSL ES 3.00 that the reference's generated tracer fragment shader uses, so that the shader
// text (never stored in this repository: oracle/gen_golden/ref_capture.js --dump-glsl writes it to a scratch directory)
// can be compiled as C++ and EXECUTED on the CPU of the build container. Fixture tooling only: nothing in the product
// path or in the GPU-side tests uses it.
//
// This is NOT a GLSL implementation and pins nothing by the rules of parity: where GLSL leaves a result to the
// implementation, this header makes the choices the CPU oracle makes (oracle/trace_oracle.c, its header comment lists
// them). What executing the text adds over reading it: control flow, indexing, operator grouping and the order of the
// floating-point operations come from the reference's own text, mechanically, and not from a transcription.
//
// Include INSIDE a namespace (the functions must hide the C library's abs / sqrt / pow / floor for unqualified calls)
// and compile with -ffp-contract=off -fno-fast-math.
//
// Choices shared with the oracle:
//   dot(a,b) = (ax*bx + ay*by) + az*bz (+ aw*bw); mat4 * vec4 adds the four column terms left to right, and the
//   translation term is dropped when w is the literal 0.0 the shader passes for directions (vec4(d, 0.0));
//   normalize(v) = v * (1 / sqrt(dot(v,v))); reflect(I,N) = I - N * (2 * dot(N,I)); mix(a,b,t) = a*(1-t) + b*t;
//   min / max drop a NaN operand (fminf / fmaxf); pow is the deterministic double-precision routine of the oracle;
//   variables without an initialiser start at zero; reads and writes outside an array are dropped (reads give zero);
//   texture() is NEAREST with CLAMP_TO_EDGE on the captured texel arrays.

typedef unsigned int uint;

static inline float g_sqrtf(float x) { return __builtin_sqrtf(x); }
static inline float g_floorf(float x) { return __builtin_floorf(x); }

struct vec4;
struct uvec4;
struct vec2 {
    union { float x, r; };
    union { float y, g; };
    vec2() : x(0), y(0) {}
    explicit vec2(float s) : x(s), y(s) {}
    vec2(float a, float b) : x(a), y(b) {}
    explicit vec2(const uvec4 &v);
    explicit vec2(const vec4 &v);
};
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(const vec4 &v);
    float &operator[](int i) { static float sink; sink = 0.0f; return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : sink)); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : 0.0f)); }
    vec3 xyz() const { return vec3(x, y, z); }
    vec3 yzx() const { return vec3(y, z, x); }
    vec3 zxy() const { return vec3(z, x, y); }
    vec3 &operator+=(const vec3 &b) { x = x + b.x; y = y + b.y; z = z + b.z; return *this; }
    vec3 &operator*=(float s) { x = x * s; y = y * s; z = z * s; return *this; }
    vec3 &operator/=(float s) { x = x / s; y = y / s; z = z / s; return *this; }
};
struct vec4 {
    float x, y, z, w;
    bool w_is_literal_zero;   // vec4(direction, 0.0): the matrix product leaves the translation column out
    vec4() : x(0), y(0), z(0), w(0), w_is_literal_zero(false) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d), w_is_literal_zero(false) {}
    vec4(const vec3 &v, float d) : x(v.x), y(v.y), z(v.z), w(d), w_is_literal_zero(d == 0.0f) {}
};
inline vec3::vec3(const vec4 &v) : x(v.x), y(v.y), z(v.z) {}
inline vec2::vec2(const vec4 &v) : x(v.x), y(v.y) {}

struct uvec4 {
    uint x, y, z, w;
    uvec4() : x(0), y(0), z(0), w(0) {}
    explicit operator uint() const { return x; }
};
inline vec2::vec2(const uvec4 &v) : x((float)v.x), y((float)v.y) {}
struct ivec4 {
    int x, y, z, w;
    ivec4() : x(0), y(0), z(0), w(0) {}
    int operator[](int i) const { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : (i == 3 ? w : 0))); }
};
struct ivec3 {
    int x, y, z;
    ivec3() : x(0), y(0), z(0) {}
    explicit ivec3(const uvec4 &v) : x((int)v.x), y((int)v.y), z((int)v.z) {}
    int operator[](int i) const { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : 0)); }
};

struct mat4 {
    float m[16];   // column-major, as gl.uniformMatrix4fv receives it
    mat4() { for (int i = 0; i < 16; i++) m[i] = 0.0f; }
};

// arrays: accesses outside the bounds are dropped (writes) or give zero (reads), as robust buffer access does
template <class T, int N>
struct garray {
    T v[N];
    T outside;
    garray() : v(), outside() {}
    T &operator[](int i) {
        if (i >= 0 && i < N) return v[i];
        outside = T();
        return outside;
    }
    const T &operator[](int i) const { return (i >= 0 && i < N) ? v[i] : outside; }
};

static inline vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator-(const vec3 &a) { return vec3(-a.x, -a.y, -a.z); }
static inline vec3 operator*(const vec3 &a, const vec3 &b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator*(const vec3 &a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator*(float s, const vec3 &a) { return vec3(s * a.x, s * a.y, s * a.z); }
static inline vec3 operator/(const vec3 &a, const vec3 &b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline vec3 operator/(const vec3 &a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }

static inline vec4 operator*(const mat4 &M, const vec4 &p) {
    const float *m = M.m;
    vec4 r;
    if (p.w_is_literal_zero) {
        r.x = (m[0] * p.x + m[4] * p.y) + m[8] * p.z;
        r.y = (m[1] * p.x + m[5] * p.y) + m[9] * p.z;
        r.z = (m[2] * p.x + m[6] * p.y) + m[10] * p.z;
        r.w = (m[3] * p.x + m[7] * p.y) + m[11] * p.z;
    } else {
        r.x = ((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12] * p.w;
        r.y = ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13] * p.w;
        r.z = ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14] * p.w;
        r.w = ((m[3] * p.x + m[7] * p.y) + m[11] * p.z) + m[15] * p.w;
    }
    return r;
}

static inline vec2 operator+(const vec2 &a, const vec2 &b) { return vec2(a.x + b.x, a.y + b.y); }

static inline float dot(const vec3 &a, const vec3 &b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline vec3 cross(const vec3 &a, const vec3 &b) {
    return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline float sqrt(float x) { return g_sqrtf(x); }
static inline float floor(float x) { return g_floorf(x); }
static inline float abs(float x) { return x < 0.0f ? -x : (x == 0.0f ? 0.0f : x); }
static inline vec3 abs(const vec3 &a) { return vec3(abs(a.x), abs(a.y), abs(a.z)); }
static inline float sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
static inline vec3 sign(const vec3 &a) { return vec3(sign(a.x), sign(a.y), sign(a.z)); }
static inline float step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
static inline vec3 step(const vec3 &e, const vec3 &a) { return vec3(step(e.x, a.x), step(e.y, a.y), step(e.z, a.z)); }
static inline float max(float a, float b) { return __builtin_fmaxf(a, b); }
static inline float min(float a, float b) { return __builtin_fminf(a, b); }
static inline float clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline vec2 clamp(const vec2 &v, const vec2 &lo, const vec2 &hi) { return vec2(clamp(v.x, lo.x, hi.x), clamp(v.y, lo.y, hi.y)); }
static inline float mod(float x, float y) { return x - y * g_floorf(x / y); }
static inline vec3 normalize(const vec3 &v) { float inv = 1.0f / g_sqrtf(dot(v, v)); return v * inv; }
static inline vec3 reflect(const vec3 &I, const vec3 &N) { return I - N * (2.0f * dot(N, I)); }
static inline vec3 mix(const vec3 &a, const vec3 &b, float t) {
    float s = 1.0f - t;
    return vec3(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t);
}

// pow(x, y), x >= 0, y > 0: the oracle's routine (double-precision atanh-series logarithm and Taylor exponential with
// fixed coefficients: only IEEE add / mul / div)
static inline float pow(float xf, float yf) {
    if (!(xf > 0.0f)) return 0.0f;
    if (xf == 1.0f) return 1.0f;
    double x = (double)xf;
    unsigned long long bits;
    __builtin_memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    __builtin_memcpy(&m, &bits, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0), s2 = s * s;
    double p = 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0; p = p * s2 + 1.0 / 19.0; p = p * s2 + 1.0 / 17.0; p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0; p = p * s2 + 1.0 / 11.0; p = p * s2 + 1.0 / 9.0;  p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;  p = p * s2 + 1.0 / 3.0;  p = p * s2 + 1.0;
    double ln_m = 2.0 * s * p;
    double lg = (double)e + ln_m * 1.4426950408889634;
    double t = (double)yf * lg;
    if (t < -160.0) return 0.0f;
    if (t > 130.0) return __builtin_inff();
    double n = __builtin_floor(t + 0.5), f = (t - n) * 0.6931471805599453;
    double q = 1.0 / 87178291200.0;
    q = q * f + 1.0 / 6227020800.0; q = q * f + 1.0 / 479001600.0; q = q * f + 1.0 / 39916800.0;
    q = q * f + 1.0 / 3628800.0;    q = q * f + 1.0 / 362880.0;    q = q * f + 1.0 / 40320.0;
    q = q * f + 1.0 / 5040.0;       q = q * f + 1.0 / 720.0;       q = q * f + 1.0 / 120.0;
    q = q * f + 1.0 / 24.0;         q = q * f + 1.0 / 6.0;         q = q * f + 0.5;
    q = q * f + 1.0;                q = q * f + 1.0;
    unsigned long long sb = (unsigned long long)((long long)n + 1023) << 52;
    double sc;
    __builtin_memcpy(&sc, &sb, 8);
    return (float)(q * sc);
}

// ---- textures: the payloads of the reference's texSubImage2D calls, NEAREST, CLAMP_TO_EDGE ---------------------------------
struct TexImage {
    const void *data;
    int width, height, channels;
    bool unorm8;   // RGBA8 render target: a byte b reads as b / 255
};
static TexImage g_units[8];

static inline void texel_of(const TexImage &t, const vec2 &c, int &tx, int &ty) {
    tx = (int)g_floorf(c.x * (float)t.width);
    ty = (int)g_floorf(c.y * (float)t.height);
    tx = tx < 0 ? 0 : (tx > t.width - 1 ? t.width - 1 : tx);
    ty = ty < 0 ? 0 : (ty > t.height - 1 ? t.height - 1 : ty);
}
struct sampler2D { int unit; sampler2D() : unit(0) {} sampler2D &operator=(int u) { unit = u; return *this; } };
struct usampler2D { int unit; usampler2D() : unit(0) {} usampler2D &operator=(int u) { unit = u; return *this; } };
struct isampler2D { int unit; isampler2D() : unit(0) {} isampler2D &operator=(int u) { unit = u; return *this; } };

static inline vec4 texture(const sampler2D &s, const vec2 &c) {
    const TexImage &t = g_units[s.unit];
    int tx, ty;
    texel_of(t, c, tx, ty);
    vec4 r(0.0f, 0.0f, 0.0f, 1.0f);
    if (t.unorm8) {
        const unsigned char *b = (const unsigned char *)t.data + ((size_t)ty * t.width + tx) * 4;
        r.x = (float)b[0] / 255.0f; r.y = (float)b[1] / 255.0f; r.z = (float)b[2] / 255.0f; r.w = (float)b[3] / 255.0f;
        return r;
    }
    const float *p = (const float *)t.data + ((size_t)ty * t.width + tx) * t.channels;
    if (t.channels > 0) r.x = p[0];
    if (t.channels > 1) r.y = p[1];
    if (t.channels > 2) r.z = p[2];
    if (t.channels > 3) r.w = p[3];
    return r;
}
static inline uvec4 texture(const usampler2D &s, const vec2 &c) {
    const TexImage &t = g_units[s.unit];
    int tx, ty;
    texel_of(t, c, tx, ty);
    const uint *p = (const uint *)t.data + ((size_t)ty * t.width + tx) * t.channels;
    uvec4 r;
    r.w = 1u;
    if (t.channels > 0) r.x = p[0];
    if (t.channels > 1) r.y = p[1];
    if (t.channels > 2) r.z = p[2];
    if (t.channels > 3) r.w = p[3];
    return r;
}
static inline ivec4 texture(const isampler2D &s, const vec2 &c) {
    const TexImage &t = g_units[s.unit];
    int tx, ty;
    texel_of(t, c, tx, ty);
    const int *p = (const int *)t.data + ((size_t)ty * t.width + tx) * t.channels;
    ivec4 r;
    r.w = 1;
    if (t.channels > 0) r.x = p[0];
    if (t.channels > 1) r.y = p[1];
    if (t.channels > 2) r.z = p[2];
    if (t.channels > 3) r.w = p[3];
    return r;
}
