/*
 * mirhi_oracle.c -- scalar CPU restatement of the reference draw path (see mirhi_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into, imported by or called from the product path.
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 * All arithmetic that decides coverage / depth / winning primitive is either integer or
 * IEEE binary32 {+,-,*,/} in a fixed order with contraction off, so a second implementation
 * that follows DESIGN.md "Pipeline specification" reproduces it bit for bit.
 *
 * Pipeline stages and the reference items they restate:
 *   a1/a2 vertex + index fetch      crates/rhi/src/vertex.rs:20-61,88-170; command.rs:583-628
 *   a4    vertex shaders            shaders/hlsl/vertex/triangle.hlsl:16-22, vertex/model.hlsl:39-68
 *   a5    clip, divide, viewport, cull   pipeline.rs:645-698,976-986; renderer.rs:504-518 (+ Vulkan 1.3 rules)
 *   a6    coverage, top-left rule   pipeline.rs:976-992
 *   a7    depth test / write        pipeline.rs:677-679,997-1004; rendering.rs:356-370; depth_buffer.rs:48
 *   a8    pixel shaders             pixel/triangle.hlsl:10-13, pixel/model.hlsl:29-82,
 *                                   pixel/model_full.hlsl:63-150, lights.hlsli:63-231
 *   a9    clear, store, sRGB        renderer.rs:479-488; swapchain.rs:561-570; pipeline.rs:499-512
 */
#include "mirhi_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * small vector helpers (binary32, no contraction)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float x, y, z, w; } v4;
typedef struct { float x, y, z; } v3;

static float rdf(const void* p, uint32_t off) { float f; memcpy(&f, (const uint8_t*)p + off, 4); return f; }
static uint32_t rdu(const void* p, uint32_t off) { uint32_t u; memcpy(&u, (const uint8_t*)p + off, 4); return u; }

/* HLSL mul(M, v) with M stored column-major (glam Mat4 bytes): r[row] = sum_c M[c][row]*v[c],
 * accumulated left to right (model.hlsl:44,48). */
static v4 mat4_mul_v4(const float* m, v4 v) {
    v4 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}
/* mul((float3x3)M, v) (model.hlsl:51-52) */
static v3 mat3_mul_v3(const float* m, v3 v) {
    v3 r;
    r.x = (m[0] * v.x + m[4] * v.y) + m[8] * v.z;
    r.y = (m[1] * v.x + m[5] * v.y) + m[9] * v.z;
    r.z = (m[2] * v.x + m[6] * v.y) + m[10] * v.z;
    return r;
}
static float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static v3 add3(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static v3 sub3(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static v3 scale3(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static v3 mul3(v3 a, v3 b) { v3 r = {a.x * b.x, a.y * b.y, a.z * b.z}; return r; }
static v3 cross3(v3 a, v3 b) {
    v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
static float length3(v3 a) { return sqrtf(dot3(a, a)); }
/* HLSL normalize = v * rsqrt(dot(v,v)); restated with a correctly rounded 1/sqrt */
static v3 normalize3(v3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); v3 r = {a.x * inv, a.y * inv, a.z * inv}; return r; }
static float saturatef(float x) { return x > 0.0f ? (x < 1.0f ? x : 1.0f) : 0.0f; }

/* ------------------------------------------------------------------------------------------------
 * a8: lights.hlsli
 * ---------------------------------------------------------------------------------------------- */
float oracle_attenuation(float distance, float radius) {       /* lights.hlsli:63-73 */
    float attenuation = 1.0f / (distance * distance + 1.0f);
    float falloff = saturatef(1.0f - distance / radius);
    falloff = falloff * falloff;
    return attenuation * falloff;
}
static float spot_attenuation(v3 lightDir, v3 spotDir, float innerCos, float outerCos) { /* :77-81 */
    v3 nl = {-lightDir.x, -lightDir.y, -lightDir.z};
    float cosAngle = dot3(nl, spotDir);
    return saturatef((cosAngle - outerCos) / (innerCos - outerCos));
}
float oracle_roughness_to_shininess(float roughness) {          /* lights.hlsli:152-159 */
    float r = roughness < 0.0f ? 0.0f : (roughness > 1.0f ? 1.0f : roughness);
    return 2048.0f + (2.0f - 2048.0f) * r;                      /* lerp(2048, 2, r) */
}
static v3 blinn_phong(v3 lightDir, v3 viewDir, v3 normal, v3 lightColor, v3 albedo, float shininess) {
    /* lights.hlsli:95-117 */
    float NdotL = dot3(normal, lightDir);
    if (!(NdotL > 0.0f)) NdotL = 0.0f;                           /* max(dot, 0) */
    v3 diffuse = mul3(scale3(lightColor, NdotL), albedo);
    if (NdotL <= 0.0f) return diffuse;
    v3 halfDir = normalize3(add3(lightDir, viewDir));
    float NdotH = dot3(normal, halfDir);
    if (!(NdotH > 0.0f)) NdotH = 0.0f;
    float sp = powf(NdotH, shininess);
    return add3(diffuse, scale3(lightColor, sp));
}
void oracle_blinn_phong(const float L[3], const float V[3], const float N[3], const float light_color[3],
                        const float albedo[3], float shininess, float out[3]) {
    v3 l = {L[0], L[1], L[2]}, v = {V[0], V[1], V[2]}, n = {N[0], N[1], N[2]};
    v3 lc = {light_color[0], light_color[1], light_color[2]}, al = {albedo[0], albedo[1], albedo[2]};
    v3 r = blinn_phong(l, v, n, lc, al, shininess);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ------------------------------------------------------------------------------------------------
 * a9: sRGB OETF + UNORM8 (swapchain.rs:561-570 selects B8G8R8A8_SRGB)
 * ---------------------------------------------------------------------------------------------- */
uint8_t oracle_srgb8(float c) {
    c = saturatef(c);
    float e = (c <= 0.0031308f) ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
    e = saturatef(e);
    return (uint8_t)rintf(e * 255.0f);
}
static uint8_t unorm8(float a) { return (uint8_t)rintf(saturatef(a) * 255.0f); }

/* ------------------------------------------------------------------------------------------------
 * texture sampling: bilinear or trilinear, repeat (model_full.hlsl:44-46 declares one linear sampler;
 * crates/rhi/src/{image,sampler,texture}.rs are empty stubs, so addressing/filtering is this
 * build's stated choice: VK_SAMPLER_ADDRESS_MODE_REPEAT, VK_FILTER_LINEAR, UNORM texels)
 * ---------------------------------------------------------------------------------------------- */
static float srgb_to_linear(uint8_t byte) {                     /* sRGB EOTF per byte: double, rounded once (as mirhi_api.hip) */
    static float lut[256];
    static int ready = 0;
    if (!ready) {
        for (int i = 0; i < 256; i++) {
            double c = (double)i / 255.0;
            lut[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
        }
        ready = 1;      /* benign race between band threads: every writer stores the same values */
    }
    return lut[byte];
}
static v4 texel(const uint8_t* level, int32_t w, int32_t h, int32_t x, int32_t y, int srgb) {
    x %= w; if (x < 0) x += w;
    y %= h; if (y < 0) y += h;
    const uint8_t* p = level + 4u * ((uint32_t)y * (uint32_t)w + (uint32_t)x);
    v4 r = {(float)p[0] * (1.0f / 255.0f), (float)p[1] * (1.0f / 255.0f), (float)p[2] * (1.0f / 255.0f),
            (float)p[3] * (1.0f / 255.0f)};
    if (srgb) { r.x = srgb_to_linear(p[0]); r.y = srgb_to_linear(p[1]); r.z = srgb_to_linear(p[2]); }
    return r;
}
/* bilinear tap of one mip level, repeat addressing */
static v4 sample_level(const uint8_t* level, uint32_t w, uint32_t h, float u, float v, int srgb) {
    float fx = u * (float)w - 0.5f, fy = v * (float)h - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float ax = fx - x0f, ay = fy - y0f;
    int32_t x0 = (int32_t)x0f, y0 = (int32_t)y0f;
    v4 c00 = texel(level, (int32_t)w, (int32_t)h, x0, y0, srgb), c10 = texel(level, (int32_t)w, (int32_t)h, x0 + 1, y0, srgb);
    v4 c01 = texel(level, (int32_t)w, (int32_t)h, x0, y0 + 1, srgb), c11 = texel(level, (int32_t)w, (int32_t)h, x0 + 1, y0 + 1, srgb);
    v4 r;
#define LERP2(f) r.f = (c00.f + (c10.f - c00.f) * ax) + ((c01.f + (c11.f - c01.f) * ax) - (c00.f + (c10.f - c00.f) * ax)) * ay
    LERP2(x); LERP2(y); LERP2(z); LERP2(w);
#undef LERP2
    return r;
}
/* screen-space derivatives of (u, v): the value one pixel to the right / one pixel down minus the value at the pixel */
typedef struct { float dudx, dvdx, dudy, dvdy; } uv_grad;
/* Without a chain: bilinear.  With one: trilinear, lambda = 0.5 * log2(max(|d(uv*size)/dx|^2, |d(uv*size)/dy|^2)) clamped to
 * [0, levels - 1] (Vulkan 1.3 15.6.7 with the exact footprint axes instead of an approximation) */
static v4 sample_trilinear(const oracle_texture* t, uint32_t levels, float u, float v, float lam) {
    if (!(lam > 0.0f)) lam = 0.0f;                              /* magnification, zero footprint, NaN */
    float top = (float)(levels - 1u);
    if (lam > top) lam = top;
    float l0f = floorf(lam), f = lam - l0f;
    uint32_t l0 = (uint32_t)l0f, lw = t->width, lh = t->height;
    const uint8_t* level = t->rgba8;
    for (uint32_t l = 0; l < l0; l++) { level += 4u * (size_t)lw * lh; lw = lw > 1u ? lw >> 1 : 1u; lh = lh > 1u ? lh >> 1 : 1u; }
    v4 c0 = sample_level(level, lw, lh, u, v, (int)t->srgb);
    if (l0 + 1u >= levels) return c0;
    v4 c1 = sample_level(level + 4u * (size_t)lw * lh, lw > 1u ? lw >> 1 : 1u, lh > 1u ? lh >> 1 : 1u, u, v, (int)t->srgb);
    v4 r = {c0.x + (c1.x - c0.x) * f, c0.y + (c1.y - c0.y) * f, c0.z + (c1.z - c0.z) * f, c0.w + (c1.w - c0.w) * f};
    return r;
}
/* max_anisotropy > 1 on a texture with a chain (the device enables `sampler_anisotropy`, crates/rhi/src/device.rs:161-165; the
 * reference never creates a sampler, so the filter is the one the Vulkan specification gives as its example, 16.8 "Texel
 * Anisotropic Filtering" / VK_EXT_texture_filter_anisotropic): with Pmax / Pmin the longer / shorter footprint axis in texels,
 * N = min(ceil(Pmax / Pmin), max_anisotropy), lambda = log2(Pmax / N), and the result is the mean of N trilinear taps placed at
 * (u, v) + (i / (N + 1) - 1/2) * d(u, v)/d(major axis), i = 1..N. */
static v4 sample_texture(const oracle_texture* t, float u, float v, const uv_grad* g) {
    if (!t->rgba8 || t->width == 0 || t->height == 0) { v4 one = {1.0f, 1.0f, 1.0f, 1.0f}; return one; }
    const uint32_t levels = t->levels ? t->levels : 1u;
    if (levels <= 1u) return sample_level(t->rgba8, t->width, t->height, u, v, (int)t->srgb);
    float ax = g->dudx * (float)t->width, bx = g->dvdx * (float)t->height, ay = g->dudy * (float)t->width, by = g->dvdy * (float)t->height;
    float rx = ax * ax + bx * bx, ry = ay * ay + by * by;
    if (t->max_anisotropy <= 1u) return sample_trilinear(t, levels, u, v, 0.5f * log2f(rx > ry ? rx : ry));
    const int major_x = rx > ry;
    float pmax = sqrtf(major_x ? rx : ry), pmin = sqrtf(major_x ? ry : rx);
    float nf = ceilf(pmax / pmin);
    if (!(nf <= (float)t->max_anisotropy)) nf = (float)t->max_anisotropy;      /* also a zero short axis (inf) and a zero footprint (NaN) */
    if (!(nf >= 1.0f)) nf = 1.0f;
    const float lam = log2f(pmax / nf);
    const float du = major_x ? g->dudx : g->dudy, dv = major_x ? g->dvdx : g->dvdy;
    v4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t n = (uint32_t)nf;
    for (uint32_t i = 1; i <= n; i++) {
        const float ti = (float)i / (nf + 1.0f) - 0.5f;
        v4 s = sample_trilinear(t, levels, u + du * ti, v + dv * ti, lam);
        acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
    }
    v4 r = {acc.x / nf, acc.y / nf, acc.z / nf, acc.w / nf};
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * a1/a2/a4: vertex fetch + vertex shader position
 * ---------------------------------------------------------------------------------------------- */
static uint32_t fetch_index(const oracle_draw* d, uint32_t k) {
    /* k-th vertex of the draw -> vertex buffer element (command.rs:583-628) */
    if (d->index_type == 0) return d->first + k;
    uint32_t idx;
    if (d->index_type == 2) { uint16_t s; memcpy(&s, (const uint8_t*)d->index_data + 2u * (size_t)(d->first + k), 2); idx = s; }
    else { memcpy(&idx, (const uint8_t*)d->index_data + 4u * (size_t)(d->first + k), 4); }
    return (uint32_t)((int32_t)idx + d->vertex_offset);
}
static v4 vs_position(const oracle_draw* d, uint32_t vidx, v3* world_out) {
    const uint8_t* v = d->vertex_data + (size_t)vidx * d->vertex_stride;
    v4 p = {rdf(v, 0), rdf(v, 4), rdf(v, 8), 1.0f};
    if (d->program == ORACLE_PROGRAM_TRIANGLE) {                 /* vertex/triangle.hlsl:19 */
        if (world_out) { world_out->x = p.x; world_out->y = p.y; world_out->z = p.z; }
        return p;
    }
    const float* model = (const float*)d->object;               /* ObjectData.model @0 */
    const float* viewproj = (const float*)d->camera + 32;       /* CameraData.viewProjection @128 */
    v4 world = mat4_mul_v4(model, p);                            /* vertex/model.hlsl:44 */
    if (world_out) { world_out->x = world.x; world_out->y = world.y; world_out->z = world.z; }
    return mat4_mul_v4(viewproj, world);                         /* vertex/model.hlsl:48 */
}

/* ------------------------------------------------------------------------------------------------
 * a5: clipping (Vulkan clip volume -w<=x,y<=w, 0<=z<=w; depth_clamp off pipeline.rs:663)
 * x/y are not clipped to the view volume (coverage is limited by scissor instead) but to a
 * guard band that keeps the snapped coordinates inside +-2^22 sub-pixels.
 * ---------------------------------------------------------------------------------------------- */
#define GUARD_PX 16000.0f
enum { PL_NEAR = 1, PL_FAR = 2, PL_GL = 4, PL_GR = 8, PL_GT = 16, PL_GB = 32 };

static float plane_dist(int plane, v4 c, float gx, float gy) {
    switch (plane) {
        case PL_NEAR: return c.z;
        case PL_FAR:  return c.w - c.z;
        case PL_GL:   return c.x + gx * c.w;
        case PL_GR:   return gx * c.w - c.x;
        case PL_GT:   return c.y + gy * c.w;
        default:      return gy * c.w - c.y;
    }
}
static uint32_t outcode_view(v4 c) {
    uint32_t oc = 0;
    if (c.x < -c.w) oc |= 1; if (c.x > c.w) oc |= 2;
    if (c.y < -c.w) oc |= 4; if (c.y > c.w) oc |= 8;
    if (c.z < 0.0f) oc |= 16; if (c.z > c.w) oc |= 32;
    return oc;
}
static uint32_t outcode_clip(v4 c, float gx, float gy) {
    uint32_t oc = 0;
    if (c.z < 0.0f) oc |= PL_NEAR;
    if (c.w - c.z < 0.0f) oc |= PL_FAR;
    if (c.x + gx * c.w < 0.0f) oc |= PL_GL;
    if (gx * c.w - c.x < 0.0f) oc |= PL_GR;
    if (c.y + gy * c.w < 0.0f) oc |= PL_GT;
    if (gy * c.w - c.y < 0.0f) oc |= PL_GB;
    return oc;
}
/* Sutherland-Hodgman against one plane; new vertices are always interpolated from the inside
 * vertex towards the outside vertex so that an edge shared by two triangles clips identically. */
static int clip_polygon(int plane, const v4* in, int n, v4* out, float gx, float gy) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        v4 a = in[i], b = in[(i + 1) % n];
        float da = plane_dist(plane, a, gx, gy), db = plane_dist(plane, b, gx, gy);
        int ina = da >= 0.0f, inb = db >= 0.0f;
        if (ina) out[m++] = a;
        if (ina != inb) {
            v4 p, q; float dp, dq;
            if (ina) { p = a; q = b; dp = da; dq = db; } else { p = b; q = a; dp = db; dq = da; }
            float t = dp / (dp - dq);
            v4 r = {p.x + t * (q.x - p.x), p.y + t * (q.y - p.y), p.z + t * (q.z - p.z), p.w + t * (q.w - p.w)};
            out[m++] = r;
        }
    }
    return m;
}

/* ------------------------------------------------------------------------------------------------
 * a5-a7: triangle setup record
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t A[3], B[3], C[3];   /* edge functions E = A*Px + B*Py + C (+bias folded into C) in 1/256 px */
    float z0, zx, zy, x0f, y0f; /* depth plane through the snapped vertices */
    int32_t minx, maxx, miny, maxy; /* inclusive pixel bbox, already clamped to scissor */
    uint32_t prim;              /* global primitive id (draw base + triangle index) */
    uint32_t draw;
} setup_tri;

typedef struct { setup_tri* v; size_t n, cap; } tri_list;

static void push_tri(tri_list* l, const setup_tri* t) {
    if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 1024; l->v = (setup_tri*)realloc(l->v, l->cap * sizeof(setup_tri)); }
    l->v[l->n++] = *t;
}

static int64_t floor_div256(int64_t a) { return a >> 8; } /* arithmetic shift = floor for negatives */

/* screen-space triangle (clip-space vertices with w>0) -> setup record; returns 0 if rejected */
static int setup_triangle(const oracle_pass* pass, const oracle_draw* d, uint32_t draw_index, uint32_t prim,
                          const v4 c[3], setup_tri* out) {
    const float hw = 0.5f * d->viewport[2], hh = 0.5f * d->viewport[3];
    const float cx = d->viewport[0] + hw, cy = d->viewport[1] + hh;
    const float dscale = d->viewport[5] - d->viewport[4], dmin = d->viewport[4];
    int32_t X[3], Y[3]; float z[3];
    for (int i = 0; i < 3; i++) {
        if (!(c[i].w > 0.0f)) return 0;
        float iw = 1.0f / c[i].w;
        float xs = (c[i].x * iw) * hw + cx;                      /* Vulkan viewport transform */
        float ys = (c[i].y * iw) * hh + cy;
        float zs = (c[i].z * iw) * dscale + dmin;
        if (!(fabsf(xs) <= 16383.0f) || !(fabsf(ys) <= 16383.0f)) return 0;
        X[i] = (int32_t)rintf(xs * 256.0f);                      /* 8 sub-pixel bits, round-half-even */
        Y[i] = (int32_t)rintf(ys * 256.0f);
        z[i] = zs;
    }
    int64_t S = (int64_t)(X[1] - X[0]) * (int64_t)(Y[2] - Y[0]) - (int64_t)(X[2] - X[0]) * (int64_t)(Y[1] - Y[0]);
    if (S == 0) return 0;
    /* Vulkan: a = -1/2 sum(x_i y_i+1 - x_i+1 y_i) = -S/2 in framebuffer coords; CCW front <=> a > 0 */
    int front = (d->front_face == ORACLE_FRONT_CCW) ? (S < 0) : (S > 0);
    if (d->cull_mode == ORACLE_CULL_FRONT_AND_BACK) return 0;
    if (d->cull_mode == ORACLE_CULL_BACK && !front) return 0;
    if (d->cull_mode == ORACLE_CULL_FRONT && front) return 0;
    if (S < 0) { /* normalise orientation: interior has E > 0 */
        int32_t t; float tz;
        t = X[1]; X[1] = X[2]; X[2] = t; t = Y[1]; Y[1] = Y[2]; Y[2] = t; tz = z[1]; z[1] = z[2]; z[2] = tz;
    }
    for (int i = 0; i < 3; i++) {
        int a = i, b = (i + 1) % 3;
        int64_t dx = X[b] - X[a], dy = Y[b] - Y[a];
        int64_t A = -dy, B = dx;
        int64_t C = -(B * (int64_t)Y[a] + A * (int64_t)X[a]);
        int topleft = (dy < 0) || (dy == 0 && dx > 0);           /* top-left fill rule */
        out->A[i] = A; out->B[i] = B; out->C[i] = C + (topleft ? 0 : -1);
    }
    /* depth plane (linear in screen space; Vulkan: z = sum b_i z_i without perspective correction) */
    const float inv256 = 1.0f / 256.0f;
    float fx1 = (float)(X[1] - X[0]) * inv256, fy1 = (float)(Y[1] - Y[0]) * inv256;
    float fx2 = (float)(X[2] - X[0]) * inv256, fy2 = (float)(Y[2] - Y[0]) * inv256;
    float area = fx1 * fy2 - fx2 * fy1;
    float dz1 = z[1] - z[0], dz2 = z[2] - z[0];
    out->zx = (dz1 * fy2 - dz2 * fy1) / area;
    out->zy = (dz2 * fx1 - dz1 * fx2) / area;
    out->z0 = z[0];
    out->x0f = (float)X[0] * inv256;
    out->y0f = (float)Y[0] * inv256;
    /* pixel bbox: pixel centre (256p+128) within [min, max] of the snapped vertices, then scissor */
    int32_t xmin = X[0] < X[1] ? X[0] : X[1]; if (X[2] < xmin) xmin = X[2];
    int32_t xmax = X[0] > X[1] ? X[0] : X[1]; if (X[2] > xmax) xmax = X[2];
    int32_t ymin = Y[0] < Y[1] ? Y[0] : Y[1]; if (Y[2] < ymin) ymin = Y[2];
    int32_t ymax = Y[0] > Y[1] ? Y[0] : Y[1]; if (Y[2] > ymax) ymax = Y[2];
    int32_t px0 = (int32_t)floor_div256((int64_t)xmin + 127), px1 = (int32_t)floor_div256((int64_t)xmax - 128);
    int32_t py0 = (int32_t)floor_div256((int64_t)ymin + 127), py1 = (int32_t)floor_div256((int64_t)ymax - 128);
    int32_t sx0 = d->scissor[0], sy0 = d->scissor[1];
    int32_t sx1 = d->scissor[0] + d->scissor[2] - 1, sy1 = d->scissor[1] + d->scissor[3] - 1;
    if (sx0 < 0) sx0 = 0; if (sy0 < 0) sy0 = 0;
    if (sx1 > (int32_t)pass->width - 1) sx1 = (int32_t)pass->width - 1;
    if (sy1 > (int32_t)pass->height - 1) sy1 = (int32_t)pass->height - 1;
    if (px0 < sx0) px0 = sx0; if (px1 > sx1) px1 = sx1;
    if (py0 < sy0) py0 = sy0; if (py1 > sy1) py1 = sy1;
    if (px0 > px1 || py0 > py1) return 0;
    out->minx = px0; out->maxx = px1; out->miny = py0; out->maxy = py1;
    out->prim = prim; out->draw = draw_index;
    return 1;
}

static void process_triangle(const oracle_pass* pass, const oracle_draw* d, uint32_t draw_index, uint32_t prim,
                             const v4 c[3], tri_list* out) {
    uint32_t o0 = outcode_view(c[0]), o1 = outcode_view(c[1]), o2 = outcode_view(c[2]);
    if (o0 & o1 & o2) return;                                     /* outside one view-volume plane */
    const float hw = 0.5f * d->viewport[2], hh = 0.5f * d->viewport[3];
    const float cx = d->viewport[0] + hw, cy = d->viewport[1] + hh;
    const float gx = (GUARD_PX - fabsf(cx)) / hw, gy = (GUARD_PX - fabsf(cy)) / hh;
    uint32_t k0 = outcode_clip(c[0], gx, gy), k1 = outcode_clip(c[1], gx, gy), k2 = outcode_clip(c[2], gx, gy);
    uint32_t any = k0 | k1 | k2;
    setup_tri st;
    if (any == 0) {
        if (setup_triangle(pass, d, draw_index, prim, c, &st)) push_tri(out, &st);
        return;
    }
    v4 bufa[12], bufb[12];
    v4* in = bufa; v4* tmp = bufb;
    in[0] = c[0]; in[1] = c[1]; in[2] = c[2];
    int n = 3;
    for (int plane = PL_NEAR; plane <= PL_GB && n >= 3; plane <<= 1) {
        if (!(any & plane)) continue;
        n = clip_polygon(plane, in, n, tmp, gx, gy);
        v4* s = in; in = tmp; tmp = s;
    }
    for (int i = 1; i + 1 < n; i++) {
        v4 t[3] = {in[0], in[i], in[i + 1]};
        if (setup_triangle(pass, d, draw_index, prim, t, &st)) push_tri(out, &st);
    }
}

/* ------------------------------------------------------------------------------------------------
 * a7: depth compare
 * ---------------------------------------------------------------------------------------------- */
static int depth_cmp(uint32_t op, float z, float stored) {
    switch (op) {
        case ORACLE_CMP_NEVER: return 0;
        case ORACLE_CMP_LESS: return z < stored;
        case ORACLE_CMP_EQUAL: return z == stored;
        case ORACLE_CMP_LESS_OR_EQUAL: return z <= stored;
        case ORACLE_CMP_GREATER: return z > stored;
        case ORACLE_CMP_NOT_EQUAL: return z != stored;
        case ORACLE_CMP_GREATER_OR_EQUAL: return z >= stored;
        default: return 1;
    }
}

/* ------------------------------------------------------------------------------------------------
 * a8: fragment shading of one visible pixel
 * ---------------------------------------------------------------------------------------------- */
typedef struct { v3 world, normal, tangent, bitangent; float u, v; v3 color; } varyings;
static void shade_pbr(const oracle_draw* d, const float b[3], const void* vvp, v3 worldPos, v3 Nv, v3 V, const void* gradp, float rgba[4]);

static void vs_varyings(const oracle_draw* d, uint32_t vidx, v4* clip, varyings* o) {
    const uint8_t* vtx = d->vertex_data + (size_t)vidx * d->vertex_stride;
    v3 world;
    *clip = vs_position(d, vidx, &world);
    memset(o, 0, sizeof *o);
    if (d->program == ORACLE_PROGRAM_TRIANGLE) {                 /* vertex/triangle.hlsl:20 */
        o->color.x = rdf(vtx, 12); o->color.y = rdf(vtx, 16); o->color.z = rdf(vtx, 20);
        return;
    }
    const float* model = (const float*)d->object;
    const float* nmat = (const float*)d->object + 16;           /* ObjectData.normalMatrix @64 */
    v3 n = {rdf(vtx, 12), rdf(vtx, 16), rdf(vtx, 20)};
    v3 t = {rdf(vtx, 32), rdf(vtx, 36), rdf(vtx, 40)};
    float tw = rdf(vtx, 44);
    v3 N = normalize3(mat3_mul_v3(nmat, n));                     /* vertex/model.hlsl:51 */
    v3 T = normalize3(mat3_mul_v3(model, t));                    /* :52 */
    T = normalize3(sub3(T, scale3(N, dot3(T, N))));              /* :55 Gram-Schmidt */
    v3 B = scale3(cross3(N, T), tw);                             /* :58 */
    o->world = world; o->normal = N; o->tangent = T; o->bitangent = B;
    o->u = rdf(vtx, 24); o->v = rdf(vtx, 28);
}

static v3 interp3(const float b[3], v3 a0, v3 a1, v3 a2) {
    v3 r = {(b[0] * a0.x + b[1] * a1.x) + b[2] * a2.x, (b[0] * a0.y + b[1] * a1.y) + b[2] * a2.y,
            (b[0] * a0.z + b[1] * a1.z) + b[2] * a2.z};
    return r;
}

static void shade_pixel(const oracle_pass* pass, const oracle_draw* d, uint32_t local_tri, uint32_t px, uint32_t py,
                        float rgba[4]) {
    (void)pass;
    v4 c[3]; varyings vv[3];
    for (int k = 0; k < 3; k++) vs_varyings(d, fetch_index(d, 3u * local_tri + (uint32_t)k), &c[k], &vv[k]);
    /* perspective-correct barycentrics of the pixel centre from the ORIGINAL clip-space triangle
     * (2-D homogeneous form, pixel-relative so it is well conditioned and valid for w<=0 vertices):
     * lambda_i ~ det[p, v_j, v_k];  f = sum(b_i f_i) is then f = sum(l_i f_i/w_i)/sum(l_i/w_i) */
    const float hw = 0.5f * d->viewport[2], hh = 0.5f * d->viewport[3];
    const float cx = d->viewport[0] + hw, cy = d->viewport[1] + hh;
    const float pxc = (float)px + 0.5f, pyc = (float)py + 0.5f;
    float ax[3], ay[3];
    for (int k = 0; k < 3; k++) {
        ax[k] = (c[k].x * hw + c[k].w * cx) - pxc * c[k].w;
        ay[k] = (c[k].y * hh + c[k].w * cy) - pyc * c[k].w;
    }
    float l0 = ax[1] * ay[2] - ax[2] * ay[1];
    float l1 = ax[2] * ay[0] - ax[0] * ay[2];
    float l2 = ax[0] * ay[1] - ax[1] * ay[0];
    float inv = 1.0f / ((l0 + l1) + l2);
    float b[3] = {l0 * inv, l1 * inv, l2 * inv};

    if (d->program == ORACLE_PROGRAM_TRIANGLE) {                 /* pixel/triangle.hlsl:10-13 */
        v3 col = interp3(b, vv[0].color, vv[1].color, vv[2].color);
        rgba[0] = col.x; rgba[1] = col.y; rgba[2] = col.z; rgba[3] = 1.0f;
        return;
    }
    v3 worldPos = interp3(b, vv[0].world, vv[1].world, vv[2].world);
    v3 Nv = interp3(b, vv[0].normal, vv[1].normal, vv[2].normal);
    v3 camPos = {rdf(d->camera, 192), rdf(d->camera, 196), rdf(d->camera, 200)};
    v3 V = normalize3(sub3(camPos, worldPos));

    if (d->program == ORACLE_PROGRAM_MODEL) {                    /* pixel/model.hlsl:29-82 */
        v3 albedo = {0.7f, 0.7f, 0.7f};
        float roughness = 0.5f, ao = 1.0f;
        v3 one = {1.0f, 1.0f, 1.0f};
        v3 lightDirection = normalize3(one);
        v3 lightColor = {1.0f, 1.0f, 1.0f};
        float lightIntensity = 1.0f;
        v3 N = normalize3(Nv);
        v3 ambient = scale3(scale3(albedo, 0.03f), ao);
        float shininess = oracle_roughness_to_shininess(roughness);
        v3 lighting = blinn_phong(lightDirection, V, N, scale3(lightColor, lightIntensity), albedo, shininess);
        v3 col = add3(ambient, lighting);
        rgba[0] = col.x; rgba[1] = col.y; rgba[2] = col.z; rgba[3] = 1.0f;
        return;
    }
    /* mip-mapped textures: footprint of (u, v) from the same interpolation one pixel right / down (exact for a planar triangle) */
    uv_grad grad = {0.0f, 0.0f, 0.0f, 0.0f};
    {
        const oracle_texture* tx[5] = {&d->albedo_map, &d->normal_map, &d->metallic_roughness_map, &d->occlusion_map, &d->emissive_map};
        int any = 0;
        for (int k = 0; k < (d->program == ORACLE_PROGRAM_MODEL_PBR ? 5 : 2); k++) any |= tx[k]->rgba8 && tx[k]->levels > 1u;
        if (any) {
            float u0 = (b[0] * vv[0].u + b[1] * vv[1].u) + b[2] * vv[2].u, v0 = (b[0] * vv[0].v + b[1] * vv[1].v) + b[2] * vv[2].v;
            for (int axis = 0; axis < 2; axis++) {
                const float qx = axis == 0 ? pxc + 1.0f : pxc, qy = axis == 0 ? pyc : pyc + 1.0f;
                float gx[3], gy[3];
                for (int k = 0; k < 3; k++) {
                    gx[k] = (c[k].x * hw + c[k].w * cx) - qx * c[k].w;
                    gy[k] = (c[k].y * hh + c[k].w * cy) - qy * c[k].w;
                }
                float m0 = gx[1] * gy[2] - gx[2] * gy[1], m1 = gx[2] * gy[0] - gx[0] * gy[2], m2 = gx[0] * gy[1] - gx[1] * gy[0];
                float minv = 1.0f / ((m0 + m1) + m2);
                float bb[3] = {m0 * minv, m1 * minv, m2 * minv};
                float uu = (bb[0] * vv[0].u + bb[1] * vv[1].u) + bb[2] * vv[2].u, vq = (bb[0] * vv[0].v + bb[1] * vv[1].v) + bb[2] * vv[2].v;
                if (axis == 0) { grad.dudx = uu - u0; grad.dvdx = vq - v0; } else { grad.dudy = uu - u0; grad.dvdy = vq - v0; }
            }
        }
    }
    if (d->program == ORACLE_PROGRAM_MODEL_PBR) { shade_pbr(d, b, vv, worldPos, Nv, V, &grad, rgba); return; }
    /* ORACLE_PROGRAM_MODEL_FULL: pixel/model_full.hlsl:85-150 */
    float u = (b[0] * vv[0].u + b[1] * vv[1].u) + b[2] * vv[2].u;
    float v = (b[0] * vv[0].v + b[1] * vv[1].v) + b[2] * vv[2].v;
    v4 baseColor = {rdf(d->material, 0), rdf(d->material, 4), rdf(d->material, 8), rdf(d->material, 12)};
    float roughness = rdf(d->material, 20), ao = rdf(d->material, 24);
    v4 albedoSample = sample_texture(&d->albedo_map, u, v, &grad);
    v3 albedo = {albedoSample.x * baseColor.x, albedoSample.y * baseColor.y, albedoSample.z * baseColor.z};
    v4 nc = sample_texture(&d->normal_map, u, v, &grad);
    v3 ncm1 = {nc.x - 1.0f, nc.y - 1.0f, nc.z - 1.0f};
    int hasNormalMap = length3(ncm1) > 0.01f;                    /* model_full.hlsl:94-95 */
    v3 N = normalize3(Nv);
    if (hasNormalMap) {                                          /* GetWorldNormal :63-83 */
        v3 ns = {nc.x * 2.0f - 1.0f, nc.y * 2.0f - 1.0f, nc.z * 2.0f - 1.0f};
        v3 T = normalize3(interp3(b, vv[0].tangent, vv[1].tangent, vv[2].tangent));
        v3 Bt = normalize3(interp3(b, vv[0].bitangent, vv[1].bitangent, vv[2].bitangent));
        /* mul(normalSample, float3x3(T,B,N)) = ns.x*T + ns.y*B + ns.z*N */
        v3 wn = add3(add3(scale3(T, ns.x), scale3(Bt, ns.y)), scale3(N, ns.z));
        N = normalize3(wn);
    }
    v3 ambient = scale3(scale3(albedo, 0.03f), ao);
    v3 lighting = {0.0f, 0.0f, 0.0f};
    float shininess = oracle_roughness_to_shininess(roughness);
    {   /* CalculateDirectionalLight lights.hlsli:166-179; HLSL DirectionalLight layout :17-23 */
        v3 dir = {rdf(d->light_ubo, 0), rdf(d->light_ubo, 4), rdf(d->light_ubo, 8)};
        float intensity = rdf(d->light_ubo, 12);
        v3 color = {rdf(d->light_ubo, 16), rdf(d->light_ubo, 20), rdf(d->light_ubo, 24)};
        v3 nd = {-dir.x, -dir.y, -dir.z};
        v3 lightDir = normalize3(nd);
        lighting = add3(lighting, blinn_phong(lightDir, V, N, scale3(color, intensity), albedo, shininess));
    }
    uint32_t numPoint = rdu(d->light_ubo, 32), numSpot = rdu(d->light_ubo, 36);
    for (uint32_t i = 0; i < numPoint; i++) {                    /* CalculatePointLight lights.hlsli:182-199 */
        const uint8_t* L = (const uint8_t*)d->point_lights + 32u * i;
        v3 pos = {rdf(L, 0), rdf(L, 4), rdf(L, 8)};
        float radius = rdf(L, 12);
        v3 color = {rdf(L, 16), rdf(L, 20), rdf(L, 24)};
        float intensity = rdf(L, 28);
        v3 lightVec = sub3(pos, worldPos);
        float dist = length3(lightVec);
        v3 lightDir = scale3(lightVec, 1.0f / dist);      /* lightVec / distance */
        float att = oracle_attenuation(dist, radius);
        v3 lightColor = scale3(scale3(color, intensity), att);
        lighting = add3(lighting, blinn_phong(lightDir, V, N, lightColor, albedo, shininess));
    }
    for (uint32_t j = 0; j < numSpot; j++) {                     /* CalculateSpotLight lights.hlsli:202-231 */
        const uint8_t* L = (const uint8_t*)d->spot_lights + 48u * j;
        v3 pos = {rdf(L, 0), rdf(L, 4), rdf(L, 8)};
        float innerCos = rdf(L, 12);
        v3 sdir = {rdf(L, 16), rdf(L, 20), rdf(L, 24)};
        float outerCos = rdf(L, 28);
        v3 color = {rdf(L, 32), rdf(L, 36), rdf(L, 40)};
        float intensity = rdf(L, 44);
        v3 lightVec = sub3(pos, worldPos);
        float dist = length3(lightVec);
        v3 lightDir = scale3(lightVec, 1.0f / dist);      /* lightVec / distance */
        float datt = oracle_attenuation(dist, 50.0f);
        float satt = spot_attenuation(lightDir, normalize3(sdir), innerCos, outerCos);
        v3 lightColor = scale3(scale3(scale3(color, intensity), datt), satt);
        lighting = add3(lighting, blinn_phong(lightDir, V, N, lightColor, albedo, shininess));
    }
    v3 col = add3(ambient, lighting);
    rgba[0] = col.x; rgba[1] = col.y; rgba[2] = col.z; rgba[3] = albedoSample.w * baseColor.w;
}

/* ------------------------------------------------------------------------------------------------
 * a8: Cook-Torrance GGX, shaders/hlsl/pbr.hlsli + pixel/model_pbr.hlsl (no shadow pass: shadow = 1)
 * ---------------------------------------------------------------------------------------------- */
#define PBR_PI 3.14159265358979323846f
#define PBR_EPSILON 0.0001f
static float max0(float x) { return x > 0.0f ? x : 0.0f; }                /* max(x, 0.0); NaN -> 0 */
float oracle_distribution_ggx(float NdotH, float roughness) {              /* pbr.hlsli:55-69 (NdotH already max(.,0)) */
    float a = roughness * roughness, a2 = a * a;
    float NdotH2 = NdotH * NdotH;
    float denom = NdotH2 * (a2 - 1.0f) + 1.0f;
    denom = (PBR_PI * denom) * denom;
    return a2 / (denom > PBR_EPSILON ? denom : PBR_EPSILON);
}
float oracle_geometry_schlick_ggx(float NdotV, float roughness) {          /* pbr.hlsli:83-93 */
    float r = roughness + 1.0f;
    float k = (r * r) / 8.0f;
    float denom = NdotV * (1.0f - k) + k;
    return NdotV / (denom > PBR_EPSILON ? denom : PBR_EPSILON);
}
typedef struct { v3 albedo; float metallic, roughness, ao; v3 emissive; } pbr_material;
static v3 pbr_direct(v3 N, v3 V, v3 L, v3 radiance, const pbr_material* m) {   /* CalculatePBRDirect pbr.hlsli:292-333 */
    v3 H = normalize3(add3(V, L));
    v3 F0 = {0.04f + (m->albedo.x - 0.04f) * m->metallic, 0.04f + (m->albedo.y - 0.04f) * m->metallic,
             0.04f + (m->albedo.z - 0.04f) * m->metallic};
    float NDF = oracle_distribution_ggx(max0(dot3(N, H)), m->roughness);
    float NdotV = max0(dot3(N, V)), NdotL = max0(dot3(N, L));
    float G = oracle_geometry_schlick_ggx(NdotV, m->roughness) * oracle_geometry_schlick_ggx(NdotL, m->roughness);
    float ct = saturatef(max0(dot3(H, V)));
    float p5 = powf(1.0f - ct, 5.0f);                                       /* FresnelSchlick :131-136 */
    v3 F = {F0.x + (1.0f - F0.x) * p5, F0.y + (1.0f - F0.y) * p5, F0.z + (1.0f - F0.z) * p5};
    float om = 1.0f - m->metallic;
    v3 kD = {(1.0f - F.x) * om, (1.0f - F.y) * om, (1.0f - F.z) * om};
    float ndg = NDF * G;
    float denominator = (4.0f * NdotV) * NdotL + PBR_EPSILON;
    v3 specular = {(ndg * F.x) / denominator, (ndg * F.y) / denominator, (ndg * F.z) / denominator};
    v3 r;
    r.x = (((kD.x * m->albedo.x) / PBR_PI + specular.x) * radiance.x) * NdotL;
    r.y = (((kD.y * m->albedo.y) / PBR_PI + specular.y) * radiance.y) * NdotL;
    r.z = (((kD.z * m->albedo.z) / PBR_PI + specular.z) * radiance.z) * NdotL;
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * raster + resolve + shade of one row band
 * ---------------------------------------------------------------------------------------------- */

static void shade_pbr(const oracle_draw* d, const float b[3], const void* vvp, v3 worldPos, v3 Nv, v3 V, const void* gradp, float rgba[4]) {
    const uv_grad grad = *(const uv_grad*)gradp;
    /* pixel/model_pbr.hlsl:159-320; MaterialData :36-59 (80 B) */
    const varyings* vv = (const varyings*)vvp;
    const void* M = d->material;
    float u = (b[0] * vv[0].u + b[1] * vv[1].u) + b[2] * vv[2].u;
    float v = (b[0] * vv[0].v + b[1] * vv[1].v) + b[2] * vv[2].v;
    v4 baseColor = {rdf(M, 0), rdf(M, 4), rdf(M, 8), rdf(M, 12)};
    float metallic = rdf(M, 16), roughness = rdf(M, 20), ao = rdf(M, 24), normalScale = rdf(M, 28);
    v3 emissive = {rdf(M, 32), rdf(M, 36), rdf(M, 40)};
    uint32_t hasBase = rdu(M, 48), hasNormal = rdu(M, 52), hasMR = rdu(M, 56), hasOcc = rdu(M, 60), hasEm = rdu(M, 64);
    if (hasBase) {
        v4 t = sample_texture(&d->albedo_map, u, v, &grad);
        baseColor.x = t.x * baseColor.x; baseColor.y = t.y * baseColor.y; baseColor.z = t.z * baseColor.z; baseColor.w = t.w * baseColor.w;
    }
    if (hasMR) { v4 t = sample_texture(&d->metallic_roughness_map, u, v, &grad); roughness = roughness * t.y; metallic = metallic * t.z; }
    if (hasOcc) { v4 t = sample_texture(&d->occlusion_map, u, v, &grad); ao = ao * t.x; }
    if (hasEm) { v4 t = sample_texture(&d->emissive_map, u, v, &grad); emissive.x *= t.x; emissive.y *= t.y; emissive.z *= t.z; }
    v3 N = normalize3(Nv);                                                  /* GetWorldNormal :124-151 */
    if (hasNormal) {
        v4 nc = sample_texture(&d->normal_map, u, v, &grad);
        v3 ncm1 = {nc.x - 1.0f, nc.y - 1.0f, nc.z - 1.0f};
        if (!(length3(ncm1) < 0.01f)) {
            v3 ns = {(nc.x * 2.0f - 1.0f) * normalScale, (nc.y * 2.0f - 1.0f) * normalScale, nc.z * 2.0f - 1.0f};
            ns = normalize3(ns);
            v3 T = normalize3(interp3(b, vv[0].tangent, vv[1].tangent, vv[2].tangent));
            v3 Bt = normalize3(interp3(b, vv[0].bitangent, vv[1].bitangent, vv[2].bitangent));
            N = normalize3(add3(add3(scale3(T, ns.x), scale3(Bt, ns.y)), scale3(N, ns.z)));
        }
    }
    pbr_material m;
    m.albedo.x = baseColor.x; m.albedo.y = baseColor.y; m.albedo.z = baseColor.z;
    m.metallic = metallic; m.roughness = roughness > 0.04f ? roughness : 0.04f;   /* ClampRoughness :476-479 */
    m.ao = ao; m.emissive = emissive;
    v3 lighting = {0.0f, 0.0f, 0.0f};
    {
        v3 dir = {rdf(d->light_ubo, 0), rdf(d->light_ubo, 4), rdf(d->light_ubo, 8)};
        float intensity = rdf(d->light_ubo, 12);
        v3 color = {rdf(d->light_ubo, 16), rdf(d->light_ubo, 20), rdf(d->light_ubo, 24)};
        v3 nd = {-dir.x, -dir.y, -dir.z};
        lighting = add3(lighting, pbr_direct(N, V, normalize3(nd), scale3(color, intensity), &m));   /* shadow = 1 */
    }
    uint32_t numPoint = d->point_lights ? rdu(d->light_ubo, 32) : 0, numSpot = d->spot_lights ? rdu(d->light_ubo, 36) : 0;
    for (uint32_t i = 0; i < numPoint; i++) {
        const uint8_t* L = (const uint8_t*)d->point_lights + 32u * i;
        v3 pos = {rdf(L, 0), rdf(L, 4), rdf(L, 8)};
        float radius = rdf(L, 12);
        v3 color = {rdf(L, 16), rdf(L, 20), rdf(L, 24)};
        float intensity = rdf(L, 28);
        v3 lightVec = sub3(pos, worldPos);
        float dist = length3(lightVec);
        v3 Ld = scale3(lightVec, 1.0f / dist);
        v3 radiance = scale3(scale3(color, intensity), oracle_attenuation(dist, radius));
        lighting = add3(lighting, pbr_direct(N, V, Ld, radiance, &m));
    }
    for (uint32_t j = 0; j < numSpot; j++) {
        const uint8_t* L = (const uint8_t*)d->spot_lights + 48u * j;
        v3 pos = {rdf(L, 0), rdf(L, 4), rdf(L, 8)};
        float innerCos = rdf(L, 12);
        v3 sdir = {rdf(L, 16), rdf(L, 20), rdf(L, 24)};
        float outerCos = rdf(L, 28);
        v3 color = {rdf(L, 32), rdf(L, 36), rdf(L, 40)};
        float intensity = rdf(L, 44);
        v3 lightVec = sub3(pos, worldPos);
        float dist = length3(lightVec);
        v3 Ld = scale3(lightVec, 1.0f / dist);
        float datt = oracle_attenuation(dist, 50.0f);
        float satt = spot_attenuation(Ld, normalize3(sdir), innerCos, outerCos);
        v3 radiance = scale3(scale3(scale3(color, intensity), datt), satt);
        lighting = add3(lighting, pbr_direct(N, V, Ld, radiance, &m));
    }
    /* CalculateHemisphereAmbient pbr.hlsli:483-492 */
    float up = N.y * 0.5f + 0.5f;
    v3 sky = {0.15f, 0.18f, 0.25f}, ground = {0.08f, 0.06f, 0.04f};
    v3 amb = {ground.x + (sky.x - ground.x) * up, ground.y + (sky.y - ground.y) * up, ground.z + (sky.z - ground.z) * up};
    float om = 1.0f - m.metallic;
    v3 ambient = scale3(scale3(mul3(amb, m.albedo), m.ao), om);
    float aol = 1.0f + (m.ao - 1.0f) * 0.5f;                                 /* lerp(1, ao, 0.5) :311 */
    lighting = scale3(lighting, aol);
    v3 col = add3(add3(ambient, lighting), m.emissive);
    rgba[0] = col.x; rgba[1] = col.y; rgba[2] = col.z; rgba[3] = baseColor.w;
}

typedef struct {
    const oracle_pass* pass;
    const tri_list* tris;
    const uint32_t* prim_base; /* per draw */
    uint32_t row0, row1;
    float* depth; uint32_t* prim; /* full-frame scratch, this band only touches its rows */
    float* out_rgba; uint8_t* out_bgra8;
    float* blend_color;         /* width*height*4, only when some draw blends: the pixel's current colour */
    uint8_t* blend_valid;       /* 1 = blend_color holds the pixel's colour; 0 = it is still the shade of prim[idx] (or the clear colour) */
    const uint8_t* alpha_test;  /* per draw: 1 = the draw's fragments are discarded one by one (model_pbr.hlsl:176-179 with a base colour texture) */
} band_job;

/* ------------------------------------------------------------------------------------------------
 * colour blending (Vulkan 1.3 section 28.1, configured by ColorBlendAttachment pipeline.rs:478-531)
 * ---------------------------------------------------------------------------------------------- */
static void blend_factor(uint32_t f, const float s[4], const float d[4], float rgb[3], float* a) {
    float one_m_da = 1.0f - d[3], sat = s[3] < one_m_da ? s[3] : one_m_da;
    switch (f) {
        case ORACLE_BF_ZERO: rgb[0] = rgb[1] = rgb[2] = 0.0f; *a = 0.0f; break;
        case ORACLE_BF_ONE: rgb[0] = rgb[1] = rgb[2] = 1.0f; *a = 1.0f; break;
        case ORACLE_BF_SRC_COLOR: rgb[0] = s[0]; rgb[1] = s[1]; rgb[2] = s[2]; *a = s[3]; break;
        case ORACLE_BF_ONE_MINUS_SRC_COLOR: rgb[0] = 1.0f - s[0]; rgb[1] = 1.0f - s[1]; rgb[2] = 1.0f - s[2]; *a = 1.0f - s[3]; break;
        case ORACLE_BF_DST_COLOR: rgb[0] = d[0]; rgb[1] = d[1]; rgb[2] = d[2]; *a = d[3]; break;
        case ORACLE_BF_ONE_MINUS_DST_COLOR: rgb[0] = 1.0f - d[0]; rgb[1] = 1.0f - d[1]; rgb[2] = 1.0f - d[2]; *a = 1.0f - d[3]; break;
        case ORACLE_BF_SRC_ALPHA: rgb[0] = rgb[1] = rgb[2] = s[3]; *a = s[3]; break;
        case ORACLE_BF_ONE_MINUS_SRC_ALPHA: rgb[0] = rgb[1] = rgb[2] = 1.0f - s[3]; *a = 1.0f - s[3]; break;
        case ORACLE_BF_DST_ALPHA: rgb[0] = rgb[1] = rgb[2] = d[3]; *a = d[3]; break;
        case ORACLE_BF_ONE_MINUS_DST_ALPHA: rgb[0] = rgb[1] = rgb[2] = one_m_da; *a = one_m_da; break;
        case ORACLE_BF_SRC_ALPHA_SATURATE: rgb[0] = rgb[1] = rgb[2] = sat; *a = 1.0f; break;
        default: rgb[0] = rgb[1] = rgb[2] = 0.0f; *a = 0.0f; break;   /* constant-colour factors: no blend constants on this path */
    }
}
static float blend_op(uint32_t op, float s, float sf, float d, float df) {
    switch (op) {
        case ORACLE_BO_SUBTRACT: return s * sf - d * df;
        case ORACLE_BO_REVERSE_SUBTRACT: return d * df - s * sf;
        case ORACLE_BO_MIN: return s < d ? s : d;
        case ORACLE_BO_MAX: return s > d ? s : d;
        default: return s * sf + d * df;
    }
}
static void blend_pixel(const oracle_draw* dr, const float s[4], float d[4]) {
    float sc[3], dc[3], sa, da, tmp;
    blend_factor(dr->src_color_factor, s, d, sc, &tmp);
    blend_factor(dr->dst_color_factor, s, d, dc, &tmp);
    float t3[3];
    blend_factor(dr->src_alpha_factor, s, d, t3, &sa);
    blend_factor(dr->dst_alpha_factor, s, d, t3, &da);
    float r[4];
    for (int k = 0; k < 3; k++) r[k] = blend_op(dr->color_op, s[k], sc[k], d[k], dc[k]);
    r[3] = blend_op(dr->alpha_op, s[3], sa, d[3], da);
    for (int k = 0; k < 4; k++) if (dr->color_write_mask & (1u << k)) d[k] = r[k];
}

static void* band_run(void* arg) {
    band_job* j = (band_job*)arg;
    const oracle_pass* pass = j->pass;
    const uint32_t W = pass->width;
    for (uint32_t y = j->row0; y < j->row1; y++)
        for (uint32_t x = 0; x < W; x++) { j->depth[(size_t)y * W + x] = pass->clear_depth; j->prim[(size_t)y * W + x] = ORACLE_NO_PRIM; }
    /* primitives in submission order (Vulkan rasterization order) */
    for (size_t ti = 0; ti < j->tris->n; ti++) {
        const setup_tri* t = &j->tris->v[ti];
        const oracle_draw* d = &pass->draws[t->draw];
        int32_t y0 = t->miny < (int32_t)j->row0 ? (int32_t)j->row0 : t->miny;
        int32_t y1 = t->maxy > (int32_t)j->row1 - 1 ? (int32_t)j->row1 - 1 : t->maxy;
        for (int32_t y = y0; y <= y1; y++) {
            const int64_t Py = 256 * (int64_t)y + 128;
            const float dy = ((float)y + 0.5f) - t->y0f;
            for (int32_t x = t->minx; x <= t->maxx; x++) {
                const int64_t Px = 256 * (int64_t)x + 128;
                if ((t->A[0] * Px + t->B[0] * Py + t->C[0]) < 0) continue;
                if ((t->A[1] * Px + t->B[1] * Py + t->C[1]) < 0) continue;
                if ((t->A[2] * Px + t->B[2] * Py + t->C[2]) < 0) continue;
                const float dx = ((float)x + 0.5f) - t->x0f;
                float z = fmaf(dy, t->zy, fmaf(dx, t->zx, t->z0));   /* two fused multiply-adds, single rounding each */
                z = z > 0.0f ? (z < 1.0f ? z : 1.0f) : 0.0f;
                size_t idx = (size_t)y * W + (size_t)x;
                int pass_test = d->depth_test ? depth_cmp(d->depth_compare, z, j->depth[idx]) : 1;
                if (!pass_test) continue;
                /* Pieces of ONE clipped triangle may overlap by a pixel after the snap to the sub-pixel grid (a sliver piece of the fan
                 * that flips its orientation, cull mode None).  Vulkan orders fragments between primitives, not inside one, and this
                 * restatement must not depend on the order its own clipper emits pieces in: under Always with depth write -- the one state
                 * where the order of two fragments of a primitive would show, in the stored depth -- the nearer of them is kept. */
                if (d->depth_test && d->depth_write && d->depth_compare == ORACLE_CMP_ALWAYS && j->prim[idx] == t->prim && !(z < j->depth[idx])) continue;
                float src[4];
                int have_src = 0;
                if (j->alpha_test[t->draw]) {
                    /* `if (baseColor.a < alphaCutoff) discard;` (pixel/model_pbr.hlsl:176-179): the fragment program runs, and a discarded
                     * fragment writes neither colour nor depth.  The program's alpha output is baseColor.a (:316-320 of this restatement). */
                    shade_pixel(pass, d, t->prim - j->prim_base[t->draw], (uint32_t)x, (uint32_t)y, src);
                    have_src = 1;
                    if (src[3] < rdf(d->material, 44)) continue;
                }
                if (d->blend_enable) {
                    /* the destination colour is needed now: resolve what the pixel shows so far, shade this fragment, blend */
                    float* dst = j->blend_color + 4 * idx;
                    if (!j->blend_valid[idx]) {
                        uint32_t p0 = j->prim[idx];
                        if (p0 == ORACLE_NO_PRIM) memcpy(dst, pass->clear_color, 4 * sizeof(float));
                        else {
                            uint32_t d0 = 0;
                            while (d0 + 1 < pass->num_draws && j->prim_base[d0 + 1] <= p0) d0++;
                            shade_pixel(pass, &pass->draws[d0], p0 - j->prim_base[d0], (uint32_t)x, (uint32_t)y, dst);
                        }
                        j->blend_valid[idx] = 1;
                    }
                    if (!have_src) shade_pixel(pass, d, t->prim - j->prim_base[t->draw], (uint32_t)x, (uint32_t)y, src);
                    blend_pixel(d, src, dst);
                } else if (j->blend_valid) {
                    j->blend_valid[idx] = 0;         /* opaque overwrite: the pixel is the shade of this primitive again */
                }
                j->prim[idx] = t->prim;
                if (d->depth_test && d->depth_write) j->depth[idx] = z;
            }
        }
    }
    /* shade */
    for (uint32_t y = j->row0; y < j->row1; y++) {
        for (uint32_t x = 0; x < W; x++) {
            size_t idx = (size_t)y * W + x;
            float rgba[4];
            uint32_t p = j->prim[idx];
            if (j->blend_valid && j->blend_valid[idx]) {
                memcpy(rgba, j->blend_color + 4 * idx, sizeof rgba);
            } else if (p == ORACLE_NO_PRIM) {
                memcpy(rgba, pass->clear_color, sizeof rgba);
            } else {
                uint32_t di = 0;
                while (di + 1 < pass->num_draws && j->prim_base[di + 1] <= p) di++;
                shade_pixel(pass, &pass->draws[di], p - j->prim_base[di], x, y, rgba);
            }
            if (j->out_rgba) memcpy(j->out_rgba + 4 * idx, rgba, sizeof rgba);
            if (j->out_bgra8) {
                uint8_t* o = j->out_bgra8 + 4 * idx;
                o[0] = oracle_srgb8(rgba[2]); o[1] = oracle_srgb8(rgba[1]); o[2] = oracle_srgb8(rgba[0]);
                o[3] = unorm8(rgba[3]);
            }
        }
    }
    return NULL;
}

int oracle_render(const oracle_pass* pass, int nthreads, float* out_rgba, uint32_t* out_prim, float* out_depth,
                  uint8_t* out_bgra8) {
    if (!pass || pass->width == 0 || pass->height == 0) return -1;
    const size_t npix = (size_t)pass->width * pass->height;
    tri_list tris = {0, 0, 0};
    uint32_t* prim_base = (uint32_t*)calloc(pass->num_draws + 1u, sizeof(uint32_t));
    uint8_t* alpha_test = (uint8_t*)calloc(pass->num_draws + 1u, 1);
    /* geometry: a1-a5 in submission order */
    uint32_t base = 0;
    for (uint32_t di = 0; di < pass->num_draws; di++) {
        const oracle_draw* d = &pass->draws[di];
        prim_base[di] = base;
        uint32_t ntri = d->count / 3u;                            /* TriangleList (pipeline.rs:655) */
        int dropped = 0;
        if (d->program == ORACLE_PROGRAM_MODEL_PBR) {
            /* pixel/model_pbr.hlsl:174-178 `if (baseColor.a < alphaCutoff) discard;`: alpha = baseColorFactor.a, or texel alpha in
             * [0,1] times it.  Where one decision covers the whole draw it is taken here (the draw is dropped, or no fragment can be
             * discarded); a draw whose texels could fall on both sides of the cutoff is discarded fragment by fragment in band_run. */
            float fa = rdf(d->material, 12), cutoff = rdf(d->material, 44);
            float lo = fa, hi = fa;
            if (rdu(d->material, 48) != 0) { lo = fa < 0.0f ? fa : 0.0f; hi = fa > 0.0f ? fa : 0.0f; }
            if (hi < cutoff) dropped = 1;
            else if (!(lo >= cutoff)) alpha_test[di] = 1;
        }
        for (uint32_t t = 0; t < ntri && !dropped; t++) {
            v4 c[3];
            for (uint32_t k = 0; k < 3; k++) c[k] = vs_position(d, fetch_index(d, 3u * t + k), NULL);
            process_triangle(pass, d, di, base + t, c, &tris);
        }
        base += ntri;
    }
    prim_base[pass->num_draws] = base;

    float* depth = out_depth ? out_depth : (float*)malloc(npix * sizeof(float));
    uint32_t* prim = out_prim ? out_prim : (uint32_t*)malloc(npix * sizeof(uint32_t));
    uint32_t r0 = 0, r1 = pass->height;
    if (pass->row_end > pass->row_begin) { r0 = pass->row_begin; r1 = pass->row_end < pass->height ? pass->row_end : pass->height; }
    if (r0 != 0 || r1 != pass->height) {
        /* rows outside the band are reported as clear */
        for (size_t i = 0; i < npix; i++) { depth[i] = pass->clear_depth; prim[i] = ORACLE_NO_PRIM; }
        if (out_rgba) for (size_t i = 0; i < npix; i++) memcpy(out_rgba + 4 * i, pass->clear_color, 16);
        if (out_bgra8) memset(out_bgra8, 0, npix * 4);
    }
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > r1 - r0) nthreads = (int)(r1 - r0);
    int any_blend = 0;
    for (uint32_t di = 0; di < pass->num_draws; di++) any_blend |= pass->draws[di].blend_enable != 0;
    float* blend_color = any_blend ? (float*)malloc(npix * 4 * sizeof(float)) : NULL;
    uint8_t* blend_valid = any_blend ? (uint8_t*)calloc(npix, 1) : NULL;
    band_job* jobs = (band_job*)calloc((size_t)nthreads, sizeof(band_job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    uint32_t rows = r1 - r0;
    for (int i = 0; i < nthreads; i++) {
        band_job* j = &jobs[i];
        j->pass = pass; j->tris = &tris; j->prim_base = prim_base;
        j->row0 = r0 + (uint32_t)(((uint64_t)rows * (uint64_t)i) / (uint64_t)nthreads);
        j->row1 = r0 + (uint32_t)(((uint64_t)rows * (uint64_t)(i + 1)) / (uint64_t)nthreads);
        j->depth = depth; j->prim = prim; j->out_rgba = out_rgba; j->out_bgra8 = out_bgra8;
        j->blend_color = blend_color; j->blend_valid = blend_valid; j->alpha_test = alpha_test;
        if (nthreads == 1) band_run(j);
        else pthread_create(&th[i], NULL, band_run, j);
    }
    if (nthreads > 1) for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    free(jobs); free(th);
    free(blend_color); free(blend_valid);
    if (!out_depth) free(depth);
    if (!out_prim) free(prim);
    free(prim_base); free(alpha_test);
    free(tris.v);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * glam 0.30.9 restatements (a3).  Column-major: m[4*col + row].
 * ---------------------------------------------------------------------------------------------- */
void oracle_glam_perspective_rh(float fovy, float aspect, float z_near, float z_far, float out[16]) {
    float s = sinf(0.5f * fovy), c = cosf(0.5f * fovy);
    float h = c / s, w = h / aspect, r = z_far / (z_near - z_far);
    memset(out, 0, 16 * sizeof(float));
    out[0] = w; out[5] = h; out[10] = r; out[11] = -1.0f; out[14] = r * z_near;
}
void oracle_glam_orthographic_rh(float l, float r, float b, float t, float n, float f, float out[16]) {
    float rw = 1.0f / (r - l), rh = 1.0f / (t - b), rd = 1.0f / (n - f);
    memset(out, 0, 16 * sizeof(float));
    out[0] = rw + rw; out[5] = rh + rh; out[10] = rd;
    out[12] = -(l + r) * rw; out[13] = -(t + b) * rh; out[14] = rd * n; out[15] = 1.0f;
}
void oracle_glam_look_at_rh(const float eye[3], const float center[3], const float up[3], float out[16]) {
    v3 e = {eye[0], eye[1], eye[2]}, c = {center[0], center[1], center[2]}, u0 = {up[0], up[1], up[2]};
    v3 f = normalize3(sub3(c, e));
    v3 s = normalize3(cross3(f, u0));
    v3 u = cross3(s, f);
    out[0] = s.x; out[1] = u.x; out[2] = -f.x; out[3] = 0.0f;
    out[4] = s.y; out[5] = u.y; out[6] = -f.y; out[7] = 0.0f;
    out[8] = s.z; out[9] = u.z; out[10] = -f.z; out[11] = 0.0f;
    out[12] = -dot3(e, s); out[13] = -dot3(e, u); out[14] = dot3(e, f); out[15] = 1.0f;
}
void oracle_glam_mat4_mul(const float a[16], const float b[16], float out[16]) {
    float r[16];
    for (int c = 0; c < 4; c++) {
        v4 col = {b[4 * c], b[4 * c + 1], b[4 * c + 2], b[4 * c + 3]};
        v4 o = mat4_mul_v4(a, col);
        r[4 * c] = o.x; r[4 * c + 1] = o.y; r[4 * c + 2] = o.z; r[4 * c + 3] = o.w;
    }
    memcpy(out, r, sizeof r);
}
void oracle_glam_from_scale_rotation_translation(const float s[3], const float q[4], const float t[3], float out[16]) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2;
    float wx = w * x2, wy = w * y2, wz = w * z2;
    out[0] = (1.0f - (yy + zz)) * s[0]; out[1] = (xy + wz) * s[0]; out[2] = (xz - wy) * s[0]; out[3] = 0.0f;
    out[4] = (xy - wz) * s[1]; out[5] = (1.0f - (xx + zz)) * s[1]; out[6] = (yz + wx) * s[1]; out[7] = 0.0f;
    out[8] = (xz + wy) * s[2]; out[9] = (yz - wx) * s[2]; out[10] = (1.0f - (xx + yy)) * s[2]; out[11] = 0.0f;
    out[12] = t[0]; out[13] = t[1]; out[14] = t[2]; out[15] = 1.0f;
}
static float m_at(const float* m, int row, int col) { return m[4 * col + row]; }
static float minor3(const float* m, int r0, int r1, int r2, int c0, int c1, int c2) {
    return m_at(m, r0, c0) * (m_at(m, r1, c1) * m_at(m, r2, c2) - m_at(m, r2, c1) * m_at(m, r1, c2)) -
           m_at(m, r0, c1) * (m_at(m, r1, c0) * m_at(m, r2, c2) - m_at(m, r2, c0) * m_at(m, r1, c2)) +
           m_at(m, r0, c2) * (m_at(m, r1, c0) * m_at(m, r2, c1) - m_at(m, r2, c0) * m_at(m, r1, c1));
}
float oracle_glam_determinant(const float m[16]) {
    return m_at(m, 0, 0) * minor3(m, 1, 2, 3, 1, 2, 3) - m_at(m, 0, 1) * minor3(m, 1, 2, 3, 0, 2, 3) +
           m_at(m, 0, 2) * minor3(m, 1, 2, 3, 0, 1, 3) - m_at(m, 0, 3) * minor3(m, 1, 2, 3, 0, 1, 2);
}
void oracle_glam_inverse(const float m[16], float out[16]) {
    float cof[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            int rr[3], cc[3], a = 0, b = 0;
            for (int i = 0; i < 4; i++) { if (i != r) rr[a++] = i; if (i != c) cc[b++] = i; }
            float mn = minor3(m, rr[0], rr[1], rr[2], cc[0], cc[1], cc[2]);
            cof[4 * c + r] = ((r + c) & 1) ? -mn : mn;           /* cofactor C[r][c] */
        }
    float det = m_at(m, 0, 0) * cof[0] + m_at(m, 0, 1) * cof[4] + m_at(m, 0, 2) * cof[8] + m_at(m, 0, 3) * cof[12];
    float inv_det = 1.0f / det;
    float res[16];
    /* inverse = adj / det, adj[r][c] = C[c][r] */
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) res[4 * c + r] = cof[4 * r + c] * inv_det;
    memcpy(out, res, sizeof res);
}
void oracle_glam_transpose(const float m[16], float out[16]) {
    float r[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r[4 * i + j] = m[4 * j + i];
    memcpy(out, r, sizeof r);
}
void oracle_glam_quat_mul_vec3(const float q[4], const float v[3], float out[3]) {
    float w = q[3];
    v3 b = {q[0], q[1], q[2]}, rhs = {v[0], v[1], v[2]};
    float b2 = dot3(b, b);
    v3 r = add3(add3(scale3(rhs, w * w - b2), scale3(b, dot3(rhs, b) * 2.0f)), scale3(cross3(b, rhs), w * 2.0f));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_camera_view_matrix(const float position[3], const float rotation[4], float out[16]) {
    /* camera.rs:110-114: forward = rotation * -Z; look_at_rh(position, position + forward, +Y) */
    const float negz[3] = {0.0f, 0.0f, -1.0f}, up[3] = {0.0f, 1.0f, 0.0f};
    float fwd[3], target[3];
    oracle_glam_quat_mul_vec3(rotation, negz, fwd);
    for (int i = 0; i < 3; i++) target[i] = position[i] + fwd[i];
    oracle_glam_look_at_rh(position, target, up, out);
}
void oracle_camera_projection_perspective(float fovy, float aspect, float z_near, float z_far, float out[16]) {
    /* camera.rs:117-137: perspective_rh then y_axis.y *= -1 (Vulkan Y flip) */
    oracle_glam_perspective_rh(fovy, aspect, z_near, z_far, out);
    out[5] *= -1.0f;
}
void oracle_normal_matrix(const float model[16], float out[16]) {
    /* ubo.rs:243-259 / transform.rs:163-179: identity if |det| < 1e-6 else inverse().transpose() */
    float det = oracle_glam_determinant(model);
    if (fabsf(det) < 1e-6f) {
        memset(out, 0, 16 * sizeof(float));
        out[0] = out[5] = out[10] = out[15] = 1.0f;
        return;
    }
    float inv[16];
    oracle_glam_inverse(model, inv);
    oracle_glam_transpose(inv, out);
}
