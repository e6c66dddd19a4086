// mirhi_shading.hip.h -- fragment programs: TRIANGLE, MODEL, MODEL_FULL, MODEL_PBR; texture sampling; sRGB pack (rows a8, a9)
// Part of the single device translation unit mirhi_kernels.hip (included inside namespace mirhi).
#ifndef MIRHI_SHADING_HIP_H
#define MIRHI_SHADING_HIP_H

// ------------------------------------------------------------------------------------------------
// a8: fragment programs.  Colour is tolerance-checked (|dRGB| < 1e-4 vs the oracle), not bit-exact, so
// this part may contract to FMA and use the 1-ulp hardware rcp / rsq / exp2 / log2.
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(fast)

__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// pow(x, y) for x >= 0 as exp2(y * log2(x)) (HLSL pow lowering); pow(0, y>0) = 0
// log2 near 1 comes from the series of ln(1+t) (t = x-1 is exact there): the hardware v_log_f32 has an absolute
// error of ~2^-22 around 1, which a Blinn-Phong exponent of up to 2048 would amplify past the 1e-4 colour bound.
__device__ __forceinline__ float flog2(float x) {
    const float t = x - 1.0f;
    const float p = t * (1.0f + t * (-0.5f + t * (0.33333334f + t * (-0.25f + t * 0.2f))));
    return fabsf(t) < 0.015625f ? p * 1.44269504089f : __builtin_amdgcn_logf(x);
}
__device__ __forceinline__ float fpow(float x, float y) { return __builtin_amdgcn_exp2f(y * flog2(x)); }
__device__ __forceinline__ f3 fnormalize3(f3 a) { const float r = frsq(dot3(a, a)); return {a.x * r, a.y * r, a.z * r}; }
#pragma clang fp contract(off)

// IEEE quotients, like the oracle's: a 1-ulp reciprocal here looks harmless (it only scales the light's colour) but a blended draw on
// top can feed the lit colour back through (1 - dst) factors and a reverse subtract -- the round-2 soak found one such scene in 160,000
// (seed 2018940 of tools/soak_fuzz.py: one pixel at 1.01e-4) while these two lines used v_rcp_f32.
__device__ __forceinline__ float attenuation(float distance, float radius) {          // lights.hlsli:63-73
    const float att = rcp_rn_nb(distance * distance + 1.0f);
    float falloff = saturatef(1.0f - div_rn_nb(distance, radius));
    falloff = falloff * falloff;
    return att * falloff;
}
__device__ __forceinline__ float roughness_to_shininess(float roughness) {             // lights.hlsli:152-159
    const float r = roughness < 0.0f ? 0.0f : (roughness > 1.0f ? 1.0f : roughness);
    return 2048.0f + (2.0f - 2048.0f) * r;
}
__device__ __forceinline__ f3 blinn_phong(f3 L, f3 V, f3 N, f3 lightColor, f3 albedo, float shininess) {  // :95-117
    float NdotL = dot3(N, L);
    if (!(NdotL > 0.0f)) NdotL = 0.0f;
    const f3 diffuse = mul3(scale3(lightColor, NdotL), albedo);
    if (NdotL <= 0.0f) return diffuse;
    const f3 H = normalize3(add3(L, V));
    float NdotH = dot3(N, H);
    if (!(NdotH > 0.0f)) NdotH = 0.0f;
    const float sp = fpow(NdotH, shininess);
    return add3(diffuse, scale3(lightColor, sp));
}

// sRGB byte -> linear, uploaded once per device (mirhi_api.hip builds it in double; the oracle builds the same table)
__device__ float g_srgb_lut[256];

__device__ __forceinline__ f4 unpack_rgba8(uint32_t p, bool srgb) {
    const float s = 1.0f / 255.0f;
    if (srgb) return {g_srgb_lut[p & 0xFF], g_srgb_lut[(p >> 8) & 0xFF], g_srgb_lut[(p >> 16) & 0xFF], (float)(p >> 24) * s};
    return {(float)(p & 0xFF) * s, (float)((p >> 8) & 0xFF) * s, (float)((p >> 16) & 0xFF) * s, (float)(p >> 24) * s};
}
// repeat addressing of one coordinate: c mod n into [0, n); a mask when n is a power of two (the usual case),
// one division otherwise.  The +1 neighbour wraps by comparison, so a bilinear tap costs two of these, not eight.
__device__ __forceinline__ int32_t wrap_coord(int32_t c, int32_t n) {
    if ((n & (n - 1)) == 0) return c & (n - 1);          // n is wave-uniform: a scalar branch
    c %= n;
    return c < 0 ? c + n : c;
}
// bilinear tap of one mip level, repeat addressing (see oracle sample_level)
__device__ __forceinline__ f4 sample_level(const uint32_t* texels, uint32_t w, uint32_t h, float u, float v, bool srgb) {
    if (w == 1 && h == 1) return unpack_rgba8(texels[0], srgb);
    const float fx = u * (float)w - 0.5f, fy = v * (float)h - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float ax = fx - x0f, ay = fy - y0f;
    const int32_t x0 = wrap_coord((int32_t)x0f, (int32_t)w), y0 = wrap_coord((int32_t)y0f, (int32_t)h);
    const int32_t x1 = x0 + 1 == (int32_t)w ? 0 : x0 + 1, y1 = y0 + 1 == (int32_t)h ? 0 : y0 + 1;
    const uint32_t r0 = (uint32_t)y0 * w, r1 = (uint32_t)y1 * w;
    const f4 c00 = unpack_rgba8(texels[r0 + (uint32_t)x0], srgb), c10 = unpack_rgba8(texels[r0 + (uint32_t)x1], srgb);
    const f4 c01 = unpack_rgba8(texels[r1 + (uint32_t)x0], srgb), c11 = unpack_rgba8(texels[r1 + (uint32_t)x1], srgb);
    f4 r;
#define MIRHI_LERP2(f) { const float top = c00.f + (c10.f - c00.f) * ax; const float bot = c01.f + (c11.f - c01.f) * ax; r.f = top + (bot - top) * ay; }
    MIRHI_LERP2(x) MIRHI_LERP2(y) MIRHI_LERP2(z) MIRHI_LERP2(w)
#undef MIRHI_LERP2
    return r;
}
// screen-space derivatives of the texture coordinates: (u, v) one pixel to the right and one pixel down, minus (u, v)
struct UvGrad { float dudx, dvdx, dudy, dvdy; };
// trilinear tap at level-of-detail lam (clamped to [0, levels - 1]) -- see oracle sample_trilinear
__device__ __forceinline__ f4 sample_trilinear(const uint32_t* texels, uint32_t w, uint32_t h, uint32_t levels, float u, float v, float lam, bool srgb) {
    if (!(lam > 0.0f)) lam = 0.0f;                                        // magnification, zero footprint, NaN
    const float top = (float)(levels - 1u);
    if (lam > top) lam = top;
    const float l0f = floorf(lam), f = lam - l0f;
    const uint32_t l0 = (uint32_t)l0f;
    uint32_t lw = w, lh = h;
    for (uint32_t l = 0; l < l0; l++) { texels += (size_t)lw * lh; lw = lw > 1u ? lw >> 1 : 1u; lh = lh > 1u ? lh >> 1 : 1u; }
    const f4 c0 = sample_level(texels, lw, lh, u, v, srgb);
    if (l0 + 1u >= levels) return c0;
    const uint32_t* next = texels + (size_t)lw * lh;
    const f4 c1 = sample_level(next, lw > 1u ? lw >> 1 : 1u, lh > 1u ? lh >> 1 : 1u, u, v, srgb);
    return {c0.x + (c1.x - c0.x) * f, c0.y + (c1.y - c0.y) * f, c0.z + (c1.z - c0.z) * f, c0.w + (c1.w - c0.w) * f};
}
// Texture slot `slot` of the draw at (u, v).  Without a mip chain: bilinear (the reference never creates a sampler --
// sampler.rs is a stub -- so VK_FILTER_LINEAR / REPEAT is this build's stated choice).  With a chain
// (mirhi_image_generate_mips): trilinear, LOD = log2 of the longer screen-space footprint axis in texels,
// lambda = 0.5 * log2(max(|d(uv*size)/dx|^2, |d(uv*size)/dy|^2)) clamped to [0, levels - 1].
// With a chain and max_anisotropy > 1 (mirhi_image_set_max_anisotropy; device.rs:161-165 enables the feature): the Vulkan
// specification's example filter -- N = min(ceil(Pmax / Pmin), max_anisotropy) trilinear taps along the longer footprint axis at
// lambda = log2(Pmax / N), averaged (oracle sample_texture).  N has to be the oracle's integer, so the square roots and the
// quotient that decide it are the compiler's IEEE expansions (the path is taken by whole draws, rarely: instructions do not matter).
template <bool MIPS>
__device__ __forceinline__ f4 sample_texture(DrawRef D, int slot, float u, float v, const UvGrad& g) {
    const uint32_t w = D.tex_w[slot], h = D.tex_h[slot], levels = D.tex_levels[slot];
    const uint32_t* texels = reinterpret_cast<const uint32_t*>(D.tex[slot]);
    if (!texels || w == 0 || h == 0) return {1.0f, 1.0f, 1.0f, 1.0f};
    const bool srgb = MIPS && ((D.tex_srgb >> slot) & 1u);      // (sRGB and mip-mapped textures live in the full-featured variant)
    if (!MIPS || levels <= 1u) return sample_level(texels, w, h, u, v, srgb);
    const float ax = g.dudx * (float)w, bx = g.dvdx * (float)h, ay = g.dudy * (float)w, by = g.dvdy * (float)h;
    const float rx = ax * ax + bx * bx, ry = ay * ay + by * by;
    const uint32_t max_aniso = ((D.tex_aniso >> (4 * slot)) & 15u) + 1u;          // wave-uniform (scalar loads)
    if (max_aniso <= 1u) return sample_trilinear(texels, w, h, levels, u, v, 0.5f * __builtin_amdgcn_logf(rx > ry ? rx : ry), srgb);   // v_log_f32 is log2
    const bool major_x = rx > ry;
    const float pmax = __builtin_sqrtf(major_x ? rx : ry), pmin = __builtin_sqrtf(major_x ? ry : rx);
    float nf = __builtin_ceilf(pmax / pmin);
    if (!(nf <= (float)max_aniso)) nf = (float)max_aniso;      // also a zero short axis (inf) and a zero footprint (NaN)
    if (!(nf >= 1.0f)) nf = 1.0f;
    const float lam = __builtin_amdgcn_logf(pmax / nf);
    const float du = major_x ? g.dudx : g.dudy, dv = major_x ? g.dvdx : g.dvdy;
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const float np1 = nf + 1.0f;
    // lanes run their own tap counts (at most 16)
#pragma unroll 1
    for (float fi = 1.0f; fi <= nf; fi += 1.0f) {
        const float ti = fi / np1 - 0.5f;
        const f4 sm = sample_trilinear(texels, w, h, levels, u + du * ti, v + dv * ti, lam, srgb);
        acc.x += sm.x; acc.y += sm.y; acc.z += sm.z; acc.w += sm.w;
    }
    return {acc.x / nf, acc.y / nf, acc.z / nf, acc.w / nf};
}

#pragma clang fp contract(off)
// Everything that feeds pow(NdotH, shininess) must match the oracle bit for bit: an exponent of up to 2048
// turns a 1-ulp difference in NdotH into a 1e-4 relative difference of the specular term.
// Shaded vertices are interpolated as they are fetched: ((b0 * a0 + b1 * a1) + b2 * a2) accumulated vertex by vertex -- the oracle's
// operation order -- so that one vertex's attributes are live at a time, not three (3 x 14 floats held 42 VGPRs and, together with
// the exact IEEE sequences, pushed the MODEL variants past their 96 registers).
__device__ __forceinline__ void acc1(float& a, float bk, float v, uint32_t k) { const float t = bk * v; a = k == 0u ? t : a + t; }
__device__ __forceinline__ void acc3(f3& a, float bk, f3 v, uint32_t k) { acc1(a.x, bk, v.x, k); acc1(a.y, bk, v.y, k); acc1(a.z, bk, v.z, k); }
// tangent and bitangent of the three shaded vertices (MODEL_FULL / MODEL_PBR layout, words 3 and 4), interpolated on demand: only
// pixels under a real normal map need them
__device__ __forceinline__ void interp_tangent_frame(DrawRef D, const uint32_t vi[3], const float b[3], f3& T, f3& Bt) {
    T = {0.0f, 0.0f, 0.0f}; Bt = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) {
        const uint4* sv = reinterpret_cast<const uint4*>(D.vs_attr) + (size_t)vi[k] * (D.vs_words - 1u) - 1;     // (sv[1..]: the attribute words)
        const uint4 w3 = sv[3], w4 = sv[4];
        acc3(T, b[k], {__uint_as_float(w3.x), __uint_as_float(w3.y), __uint_as_float(w3.z)}, k);
        acc3(Bt, b[k], {__uint_as_float(w3.w), __uint_as_float(w4.x), __uint_as_float(w4.y)}, k);
    }
}

__device__ __forceinline__ f3 interp3(const float b[3], f3 a0, f3 a1, f3 a2) {
    return {(b[0] * a0.x + b[1] * a1.x) + b[2] * a2.x, (b[0] * a0.y + b[1] * a1.y) + b[2] * a2.y,
            (b[0] * a0.z + b[1] * a1.z) + b[2] * a2.z};
}

// perspective-correct barycentrics of the pixel centre from the original clip-space triangle
// (2-D homogeneous form relative to the pixel: valid for w <= 0 vertices, no clipped attributes needed)
template <bool FAST>
__device__ __forceinline__ void barycentrics(DrawRef D, const f4 c[3], float pxc, float pyc, float b[3]) {
    float ax[3], ay[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        ax[k] = (c[k].x * D.hw + c[k].w * D.cx) - pxc * c[k].w;
        ay[k] = (c[k].y * D.hh + c[k].w * D.cy) - pyc * c[k].w;
    }
    const float l0 = ax[1] * ay[2] - ax[2] * ay[1];
    const float l1 = ax[2] * ay[0] - ax[0] * ay[2];
    const float l2 = ax[0] * ay[1] - ax[1] * ay[0];
    const float inv = FAST ? __builtin_amdgcn_rcpf((l0 + l1) + l2) : rcp_rn_nb((l0 + l1) + l2);
    b[0] = l0 * inv; b[1] = l1 * inv; b[2] = l2 * inv;
}

// vertex/triangle.hlsl + pixel/triangle.hlsl: clip = (pos, 1), colour pass-through.  Strict arithmetic in the oracle's
// operation order: for a sliver triangle the three lambdas nearly cancel, and a contracted FMA or the 1-ulp rcp then shows
// up as 1e-3 in the interpolated colour (found by the 60000-scene soak of tools/soak_fuzz.py; one pixel in ~3000 scenes).
// With w = 1 the homogeneous form is ax_k = (x_k * W/2 + cx) - px.  Flat-coloured triangles never get here (flat_color).
// vin: the triangle's three vertex indices (fetch_triangle_indices), which a caller that shades the same triangle several times
// fetches once
__device__ __forceinline__ f4 shade_triangle_program(DrawRef D, const uint32_t vin[3], float pxc, float pyc) {
    float ax[3], ay[3]; f3 col[3];
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) {
        const uint8_t* v = D.vb + (size_t)vin[k] * D.stride;
        ax[k] = (ldf(v, 0) * D.hw + D.cx) - pxc;                             // vertex/triangle.hlsl:19-20
        ay[k] = (ldf(v, 4) * D.hh + D.cy) - pyc;
        col[k] = {ldf(v, 12), ldf(v, 16), ldf(v, 20)};
    }
    const float l0 = ax[1] * ay[2] - ax[2] * ay[1];
    const float l1 = ax[2] * ay[0] - ax[0] * ay[2];
    const float l2 = ax[0] * ay[1] - ax[1] * ay[0];
    const float inv = rcp_rn_nb((l0 + l1) + l2);
    const float b[3] = {l0 * inv, l1 * inv, l2 * inv};
    const f3 o = interp3(b, col[0], col[1], col[2]);                         // pixel/triangle.hlsl:10-13
    return {o.x, o.y, o.z, 1.0f};
}
__device__ __forceinline__ void fetch_triangle_indices(DrawRef D, uint32_t tri, uint32_t vin[3]) {
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) vin[k] = fetch_index(D, 3u * tri + k);
}
__device__ __forceinline__ f4 shade_triangle_program(DrawRef D, uint32_t tri, float pxc, float pyc) {
    uint32_t vin[3];
    fetch_triangle_indices(D, tri, vin);
    return shade_triangle_program(D, vin, pxc, pyc);
}

// a8: Cook-Torrance GGX, shaders/hlsl/pbr.hlsli (shadow pass not on the path: shadow = 1)
#define PBR_PI 3.14159265358979323846f
#define PBR_EPSILON 0.0001f
__device__ __forceinline__ float max0(float x) { return x > 0.0f ? x : 0.0f; }
__device__ __forceinline__ float distribution_ggx(float NdotH, float roughness) {       // pbr.hlsli:55-69
    const float a = roughness * roughness, a2 = a * a;
    const float NdotH2 = NdotH * NdotH;
    float denom = NdotH2 * (a2 - 1.0f) + 1.0f;
    denom = (PBR_PI * denom) * denom;
    return div_rn_nb(a2, denom > PBR_EPSILON ? denom : PBR_EPSILON);
}
__device__ __forceinline__ float geometry_schlick_ggx(float NdotV, float roughness) {   // pbr.hlsli:83-93
    const float r = roughness + 1.0f;
    const float k = (r * r) * 0.125f;              // (exact: a power of two)
    const float denom = NdotV * (1.0f - k) + k;
    return div_rn_nb(NdotV, denom > PBR_EPSILON ? denom : PBR_EPSILON);
}
struct PbrMaterial { f3 albedo; float metallic, roughness; };
__device__ __forceinline__ f3 pbr_direct(f3 N, f3 V, f3 L, f3 radiance, const PbrMaterial& m) {   // pbr.hlsli:292-333
    const f3 H = normalize3(add3(V, L));
    const f3 F0 = {0.04f + (m.albedo.x - 0.04f) * m.metallic, 0.04f + (m.albedo.y - 0.04f) * m.metallic,
                   0.04f + (m.albedo.z - 0.04f) * m.metallic};
    const float NDF = distribution_ggx(max0(dot3(N, H)), m.roughness);
    const float NdotV = max0(dot3(N, V)), NdotL = max0(dot3(N, L));
    const float G = geometry_schlick_ggx(NdotV, m.roughness) * geometry_schlick_ggx(NdotL, m.roughness);
    const float ct = saturatef(max0(dot3(H, V)));
    const float p5 = fpow(1.0f - ct, 5.0f);                                             // FresnelSchlick :131-136
    const f3 F = {F0.x + (1.0f - F0.x) * p5, F0.y + (1.0f - F0.y) * p5, F0.z + (1.0f - F0.z) * p5};
    const float om = 1.0f - m.metallic;
    const f3 kD = {(1.0f - F.x) * om, (1.0f - F.y) * om, (1.0f - F.z) * om};
    const float ndg = NDF * G;
    const float denominator = (4.0f * NdotV) * NdotL + PBR_EPSILON;
    const f3 specular = {div_rn_nb(ndg * F.x, denominator), div_rn_nb(ndg * F.y, denominator), div_rn_nb(ndg * F.z, denominator)};
    return {((div_rn_nb(kD.x * m.albedo.x, PBR_PI) + specular.x) * radiance.x) * NdotL,
            ((div_rn_nb(kD.y * m.albedo.y, PBR_PI) + specular.y) * radiance.y) * NdotL,
            ((div_rn_nb(kD.z * m.albedo.z, PBR_PI) + specular.z) * radiance.z) * NdotL};
}

// alpha the Cook-Torrance program compares with alphaCutoff (pixel/model_pbr.hlsl:166-179) at one pixel of a draw with a base colour
// texture: the same barycentrics, (u, v), footprint and lookup as shade_model_program / shade_pbr, so the same bits
__device__ __forceinline__ float pbr_base_alpha(DrawRef D, const f4 c[3], const float uvk[3][2], float pxc, float pyc) {
    float b[3];
    barycentrics<false>(D, c, pxc, pyc, b);
    float u = 0.0f, v = 0.0f;
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) { acc1(u, b[k], uvk[k][0], k); acc1(v, b[k], uvk[k][1], k); }
    UvGrad grad = {0.0f, 0.0f, 0.0f, 0.0f};
    if (D.tex_any_mips) {
        float bx[3], by[3];
        barycentrics<false>(D, c, pxc + 1.0f, pyc, bx);
        barycentrics<false>(D, c, pxc, pyc + 1.0f, by);
        grad.dudx = ((bx[0] * uvk[0][0] + bx[1] * uvk[1][0]) + bx[2] * uvk[2][0]) - u;
        grad.dvdx = ((bx[0] * uvk[0][1] + bx[1] * uvk[1][1]) + bx[2] * uvk[2][1]) - v;
        grad.dudy = ((by[0] * uvk[0][0] + by[1] * uvk[1][0]) + by[2] * uvk[2][0]) - u;
        grad.dvdy = ((by[0] * uvk[0][1] + by[1] * uvk[1][1]) + by[2] * uvk[2][1]) - v;
    }
    return sample_texture<true>(D, 0, u, v, grad).w * ldcf(cb(D.material), 12);
}
// true if the draw's fragments have to be alpha-tested one by one: MODEL_PBR with a base colour texture whose texel alpha (in [0, 1]
// times baseColorFactor.a) can fall on both sides of alphaCutoff -- the decision geometry_body takes per draw otherwise
__device__ __forceinline__ bool draw_needs_alpha_test(DrawRef D) {
    if (D.program != 3u) return false;
    const CBytePtr M = cb(D.material);
    const float fa = ldcf(M, 12), cutoff = ldcf(M, 44);
    if (ldcu(M, 48) == 0u) return false;
    const float lo = fa < 0.0f ? fa : 0.0f, hi = fa > 0.0f ? fa : 0.0f;
    return !(hi < cutoff) && !(lo >= cutoff);
}

// pixel/model_pbr.hlsl:159-320 after the shared varying interpolation
__device__ __forceinline__ f4 shade_pbr(DrawRef D, const float b[3], const uint32_t vi[3], float u, float v, f3 worldPos, f3 V, f3 N, const UvGrad& grad) {
    const CBytePtr M = cb(D.material);                                                  // MaterialData :36-59 (80 B)
    f4 baseColor = {ldcf(M, 0), ldcf(M, 4), ldcf(M, 8), ldcf(M, 12)};
    float metallic = ldcf(M, 16), roughness = ldcf(M, 20), ao = ldcf(M, 24);
    const float normalScale = ldcf(M, 28);
    f3 emissive = {ldcf(M, 32), ldcf(M, 36), ldcf(M, 40)};
    if (ldcu(M, 48) != 0u) {
        const f4 t = sample_texture<true>(D, 0, u, v, grad);
        baseColor = {t.x * baseColor.x, t.y * baseColor.y, t.z * baseColor.z, t.w * baseColor.w};
    }
    if (ldcu(M, 56) != 0u) {
        const f4 t = sample_texture<true>(D, 2, u, v, grad);
        roughness = roughness * t.y; metallic = metallic * t.z;
    }
    if (ldcu(M, 60) != 0u) ao = ao * sample_texture<true>(D, 3, u, v, grad).x;
    if (ldcu(M, 64) != 0u) {
        const f4 t = sample_texture<true>(D, 4, u, v, grad);
        emissive = {emissive.x * t.x, emissive.y * t.y, emissive.z * t.z};
    }
    if (ldcu(M, 52) != 0u) {                                                            // GetWorldNormal :124-151
        const f4 nc = sample_texture<true>(D, 1, u, v, grad);
        const f3 ncm1 = {nc.x - 1.0f, nc.y - 1.0f, nc.z - 1.0f};
        if (!(length3(ncm1) < 0.01f)) {
            const f3 ns = normalize3({(nc.x * 2.0f - 1.0f) * normalScale, (nc.y * 2.0f - 1.0f) * normalScale, nc.z * 2.0f - 1.0f});
            f3 Ti, Bi;
            interp_tangent_frame(D, vi, b, Ti, Bi);
            const f3 T = normalize3(Ti), Bt = normalize3(Bi);
            N = normalize3(add3(add3(scale3(T, ns.x), scale3(Bt, ns.y)), scale3(N, ns.z)));
        }
    }
    PbrMaterial m;
    m.albedo = {baseColor.x, baseColor.y, baseColor.z};
    m.metallic = metallic;
    m.roughness = roughness > 0.04f ? roughness : 0.04f;                                // ClampRoughness :476-479
    f3 lighting = {0.0f, 0.0f, 0.0f};
    {
        const f3 dir = {ldcf(cb(D.lights), 0), ldcf(cb(D.lights), 4), ldcf(cb(D.lights), 8)};
        const float intensity = ldcf(cb(D.lights), 12);
        const f3 color = {ldcf(cb(D.lights), 16), ldcf(cb(D.lights), 20), ldcf(cb(D.lights), 24)};
        if (intensity != 0.0f) lighting = add3(lighting, pbr_direct(N, V, normalize3({-dir.x, -dir.y, -dir.z}), scale3(color, intensity), m));
    }
    const uint32_t numPoint = D.point_lights ? ldcu(cb(D.lights), 32) : 0u;
    const uint32_t numSpot = D.spot_lights ? ldcu(cb(D.lights), 36) : 0u;
    for (uint32_t i = 0; i < numPoint; i++) {
        const CBytePtr Lp = cb(D.point_lights) + 32u * i;
        const f3 pos = {ldcf(Lp, 0), ldcf(Lp, 4), ldcf(Lp, 8)};
        const float radius = ldcf(Lp, 12);
        const f3 color = {ldcf(Lp, 16), ldcf(Lp, 20), ldcf(Lp, 24)};
        const float intensity = ldcf(Lp, 28);
        const f3 lv = sub3(pos, worldPos);
        const float dist = length3(lv);
        const float att = attenuation(dist, radius);
        if (__ballot(att > 0.0f) == 0ull) continue;                          // out of the light's reach: exact zeros (see model_full)
        const f3 L = scale3(lv, rcp_rn_nb(dist));
        lighting = add3(lighting, pbr_direct(N, V, L, scale3(scale3(color, intensity), att), m));
    }
    for (uint32_t j = 0; j < numSpot; j++) {
        const CBytePtr Ls = cb(D.spot_lights) + 48u * j;
        const f3 pos = {ldcf(Ls, 0), ldcf(Ls, 4), ldcf(Ls, 8)};
        const float innerCos = ldcf(Ls, 12);
        const f3 sdir = {ldcf(Ls, 16), ldcf(Ls, 20), ldcf(Ls, 24)};
        const float outerCos = ldcf(Ls, 28);
        const f3 color = {ldcf(Ls, 32), ldcf(Ls, 36), ldcf(Ls, 40)};
        const float intensity = ldcf(Ls, 44);
        const f3 lv = sub3(pos, worldPos);
        const float dist = length3(lv);
        const f3 L = scale3(lv, rcp_rn_nb(dist));
        const float datt = attenuation(dist, 50.0f);
        const f3 sd = normalize3(sdir);
        const float cosAngle = dot3({-L.x, -L.y, -L.z}, sd);
        const float satt = saturatef(div_rn_nb(cosAngle - outerCos, innerCos - outerCos));
        if (__ballot(datt * satt > 0.0f) == 0ull) continue;
        lighting = add3(lighting, pbr_direct(N, V, L, scale3(scale3(scale3(color, intensity), datt), satt), m));
    }
    const float up = N.y * 0.5f + 0.5f;                                                 // CalculateHemisphereAmbient pbr.hlsli:483-492
    const f3 amb = {0.08f + (0.15f - 0.08f) * up, 0.06f + (0.18f - 0.06f) * up, 0.04f + (0.25f - 0.04f) * up};
    const float om = 1.0f - m.metallic;
    const f3 ambient = scale3(scale3(mul3(amb, m.albedo), ao), om);
    lighting = scale3(lighting, 1.0f + (ao - 1.0f) * 0.5f);                             // lerp(1, ao, 0.5) :311
    const f3 col = add3(add3(ambient, lighting), emissive);
    return {col.x, col.y, col.z, baseColor.w};
}

// FULL: the variant that also carries the Cook-Torrance program and mip-mapped (trilinear) sampling
template <bool FULL>
__device__ __forceinline__ f4 shade_model_program(DrawRef D, const uint32_t vi[3], float pxc, float pyc) {
    f4 c[3];
    const bool full = D.program >= 2;
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) {
        // vertex/model.hlsl outputs, computed once per vertex by vertex_kernel: the clip position first
        const uint4 w0 = reinterpret_cast<const uint4*>(D.vs_out)[vi[k]];
        c[k] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w)};
    }
    float b[3];
    barycentrics<false>(D, c, pxc, pyc, b);
    f3 worldPos = {0.0f, 0.0f, 0.0f}, Nv = {0.0f, 0.0f, 0.0f};
    float u = 0.0f, v = 0.0f, uvk[3][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) {
        const uint4* sv = reinterpret_cast<const uint4*>(D.vs_attr) + (size_t)vi[k] * (D.vs_words - 1u) - 1;     // (sv[1..]: the attribute words)
        const uint4 w1 = sv[1], w2 = sv[2];
        acc3(worldPos, b[k], {__uint_as_float(w1.x), __uint_as_float(w1.y), __uint_as_float(w1.z)}, k);
        acc3(Nv, b[k], {__uint_as_float(w1.w), __uint_as_float(w2.x), __uint_as_float(w2.y)}, k);
        if (full) {
            uvk[k][0] = __uint_as_float(w2.z); uvk[k][1] = __uint_as_float(w2.w);
            acc1(u, b[k], uvk[k][0], k); acc1(v, b[k], uvk[k][1], k);
        }
    }
    const CFloatPtr cam = cf(D.camera);
    const f3 camPos = {cam[48], cam[49], cam[50]};          // cameraPosition @192 B
    const f3 V = normalize3(sub3(camPos, worldPos));
    f3 N = normalize3(Nv);

    if (!full) {                                                             // pixel/model.hlsl:29-82
        const f3 albedo = {0.7f, 0.7f, 0.7f};
        const f3 one = {1.0f, 1.0f, 1.0f};
        // normalize((1, 1, 1)) is a constant: RN(1 / RN(sqrt(3))) = 0x1.279a740000000p-1 (the oracle's two roundings), not recomputed per pixel
        const float inv_len = 0x1.279a740000000p-1f;
        const f3 L = {1.0f * inv_len, 1.0f * inv_len, 1.0f * inv_len};
        const f3 ambient = scale3(scale3(albedo, 0.03f), 1.0f);
        const f3 lighting = blinn_phong(L, V, N, one, albedo, roughness_to_shininess(0.5f));
        const f3 col = add3(ambient, lighting);
        return {col.x, col.y, col.z, 1.0f};
    }
    // Mip-mapped textures need the screen-space footprint of (u, v): the same perspective-correct interpolation one pixel
    // to the right and one pixel down (exact for a planar triangle; no quad, no neighbour lane needed)
    UvGrad grad = {0.0f, 0.0f, 0.0f, 0.0f};
    if (FULL && D.tex_any_mips) {
        float bx[3], by[3];
        barycentrics<false>(D, c, pxc + 1.0f, pyc, bx);
        barycentrics<false>(D, c, pxc, pyc + 1.0f, by);
        grad.dudx = ((bx[0] * uvk[0][0] + bx[1] * uvk[1][0]) + bx[2] * uvk[2][0]) - u;
        grad.dvdx = ((bx[0] * uvk[0][1] + bx[1] * uvk[1][1]) + bx[2] * uvk[2][1]) - v;
        grad.dudy = ((by[0] * uvk[0][0] + by[1] * uvk[1][0]) + by[2] * uvk[2][0]) - u;
        grad.dvdy = ((by[0] * uvk[0][1] + by[1] * uvk[1][1]) + by[2] * uvk[2][1]) - v;
    }
    if (FULL && D.program == 3) return shade_pbr(D, b, vi, u, v, worldPos, V, N, grad);
    // pixel/model_full.hlsl:85-150
    const f4 baseColor = {ldcf(cb(D.material), 0), ldcf(cb(D.material), 4), ldcf(cb(D.material), 8), ldcf(cb(D.material), 12)};
    const float roughness = ldcf(cb(D.material), 20), ao = ldcf(cb(D.material), 24);
    const f4 albedoSample = sample_texture<FULL>(D, 0, u, v, grad);
    const f3 albedo = {albedoSample.x * baseColor.x, albedoSample.y * baseColor.y, albedoSample.z * baseColor.z};
    const f4 nc = sample_texture<FULL>(D, 1, u, v, grad);
    const f3 ncm1 = {nc.x - 1.0f, nc.y - 1.0f, nc.z - 1.0f};
    const bool hasNormalMap = length3(ncm1) > 0.01f;                          // :94-95
    if (hasNormalMap) {                                                      // GetWorldNormal :63-83
        const f3 ns = {nc.x * 2.0f - 1.0f, nc.y * 2.0f - 1.0f, nc.z * 2.0f - 1.0f};
        f3 Ti, Bi;
        interp_tangent_frame(D, vi, b, Ti, Bi);
        const f3 T = normalize3(Ti), Bt = normalize3(Bi);
        N = normalize3(add3(add3(scale3(T, ns.x), scale3(Bt, ns.y)), scale3(N, ns.z)));
    }
    const f3 ambient = scale3(scale3(albedo, 0.03f), ao);
    f3 lighting = {0.0f, 0.0f, 0.0f};
    const float shininess = roughness_to_shininess(roughness);
    {   // CalculateDirectionalLight lights.hlsli:166-179 (HLSL DirectionalLight layout :17-23)
        const f3 dir = {ldcf(cb(D.lights), 0), ldcf(cb(D.lights), 4), ldcf(cb(D.lights), 8)};
        const float intensity = ldcf(cb(D.lights), 12);
        const f3 color = {ldcf(cb(D.lights), 16), ldcf(cb(D.lights), 20), ldcf(cb(D.lights), 24)};
        // (a light of intensity 0 adds exactly +0 to every channel: skipped, wave-uniformly -- BASELINE's Phong configs run without
        // a directional light)
        if (intensity != 0.0f) {
            const f3 L = normalize3({-dir.x, -dir.y, -dir.z});
            lighting = add3(lighting, blinn_phong(L, V, N, scale3(color, intensity), albedo, shininess));
        }
    }
    const uint32_t numPoint = D.point_lights ? ldcu(cb(D.lights), 32) : 0u;
    const uint32_t numSpot = D.spot_lights ? ldcu(cb(D.lights), 36) : 0u;
    for (uint32_t i = 0; i < numPoint; i++) {                                // CalculatePointLight :182-199
        const CBytePtr Lp = cb(D.point_lights) + 32u * i;
        const f3 pos = {ldcf(Lp, 0), ldcf(Lp, 4), ldcf(Lp, 8)};
        const float radius = ldcf(Lp, 12);
        const f3 color = {ldcf(Lp, 16), ldcf(Lp, 20), ldcf(Lp, 24)};
        const float intensity = ldcf(Lp, 28);
        const f3 lv = sub3(pos, worldPos);
        const float dist = length3(lv);
        // beyond its radius the light's falloff is exactly 0, and so is everything it adds: a wave none of whose pixels the light
        // reaches skips the rest of the iteration (the other lanes of a wave that goes on add their exact zeros)
        const float att = attenuation(dist, radius);
        if (__ballot(att > 0.0f) == 0ull) continue;
        const f3 L = scale3(lv, rcp_rn_nb(dist));
        const f3 lc = scale3(scale3(color, intensity), att);
        lighting = add3(lighting, blinn_phong(L, V, N, lc, albedo, shininess));
    }
    for (uint32_t j = 0; j < numSpot; j++) {                                 // CalculateSpotLight :202-231
        const CBytePtr Ls = cb(D.spot_lights) + 48u * j;
        const f3 pos = {ldcf(Ls, 0), ldcf(Ls, 4), ldcf(Ls, 8)};
        const float innerCos = ldcf(Ls, 12);
        const f3 sdir = {ldcf(Ls, 16), ldcf(Ls, 20), ldcf(Ls, 24)};
        const float outerCos = ldcf(Ls, 28);
        const f3 color = {ldcf(Ls, 32), ldcf(Ls, 36), ldcf(Ls, 40)};
        const float intensity = ldcf(Ls, 44);
        const f3 lv = sub3(pos, worldPos);
        const float dist = length3(lv);
        const f3 L = scale3(lv, rcp_rn_nb(dist));
        const float datt = attenuation(dist, 50.0f);
        const f3 sd = normalize3(sdir);
        const float cosAngle = dot3({-L.x, -L.y, -L.z}, sd);                // CalculateSpotAttenuation :77-81
        const float satt = saturatef(div_rn_nb(cosAngle - outerCos, innerCos - outerCos));
        if (__ballot(datt * satt > 0.0f) == 0ull) continue;                  // outside the cone or the range: exact zeros, as above
        const f3 lc = scale3(scale3(scale3(color, intensity), datt), satt);
        lighting = add3(lighting, blinn_phong(L, V, N, lc, albedo, shininess));
    }
    const f3 col = add3(ambient, lighting);
    return {col.x, col.y, col.z, albedoSample.w * baseColor.w};
}

template <bool FULL>
__device__ __forceinline__ f4 shade_model_program(DrawRef D, uint32_t tri, float pxc, float pyc) {
    uint32_t vin[3];
    fetch_triangle_indices(D, tri, vin);
    return shade_model_program<FULL>(D, vin, pxc, pyc);
}

#pragma clang fp contract(fast)
// a9: sRGB OETF + UNORM8, BGRA byte order (swapchain.rs:561-570)
__device__ __forceinline__ uint32_t srgb8(float c) {
    c = saturatef(c);
    float e = (c <= 0.0031308f) ? 12.92f * c : 1.055f * __builtin_amdgcn_exp2f((1.0f / 2.4f) * __builtin_amdgcn_logf(c)) - 0.055f;
    e = saturatef(e);
    return (uint32_t)rintf(e * 255.0f);
}
__device__ __forceinline__ uint32_t pack_bgra8_srgb(f4 c) {
    return srgb8(c.z) | (srgb8(c.y) << 8) | (srgb8(c.x) << 16) | ((uint32_t)rintf(saturatef(c.w) * 255.0f) << 24);
}

#endif  // MIRHI_SHADING_HIP_H
