// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp).
// orc_texture.hpp: src/texture/{mod,mapping,uv,checkerboard,image}.rs and src/mipmap.rs (MIPMap::new, new_custom, trilinear lookup).
//
// PARITY UNPINNED for pyramid levels >= 1 of MIPMap::new: the reference builds them with the third-party `resize` crate
// (Cargo.lock: resize 0.4.3, Type::Triangle, mipmap.rs:119-128), which is not vendored under /root/reference.  resize_triangle()
// below restates that crate's published algorithm (separable, filter support scaled by the ratio when shrinking, coefficient
// lines clamped to the image and normalised, vertical pass first into a transposed f32 buffer, then the horizontal pass).  The
// reference holds no golden vectors for it; the only reference test on this path (mipmap.rs:370-388, a constant image read back
// within 6 ulps at every width) is reproduced in tests/test_textures.py.  Level 0 and everything after the pyramid is pinned by
// the reference's own source.
#pragma once
#include "orc_shapes.hpp"
#include <vector>

namespace orc {

struct MipLevel { int w = 0, h = 0; std::vector<Spectrum> data; };          // BlockedArray<T> read as level[(s, t)] = data[t*w + s]

inline Float resize_triangle_kernel(Float x) { return fmaxf(1.0f - fabsf(x), 0.0f); }
struct CoeffsLine { size_t start; std::vector<Float> data; };
inline std::vector<CoeffsLine> resize_calc_coeffs(size_t s1, size_t s2) {    // resize 0.4.3 Resizer::calc_coeffs, support = 1.0
    Float ratio = (Float)s1 / (Float)s2;
    Float filter_scale = ratio > 1.0f ? ratio : 1.0f;
    Float filter_radius = ceilf(1.0f * filter_scale);
    std::vector<CoeffsLine> out(s2);
    for (size_t x2 = 0; x2 < s2; x2++) {
        Float x1 = ((Float)x2 + 0.5f) * ratio - 0.5f;
        int64_t start = (int64_t)ceilf(x1 - filter_radius), end = (int64_t)floorf(x1 + filter_radius);
        start = std::min<int64_t>(std::max<int64_t>(start, 0), (int64_t)s1 - 1);
        end = std::min<int64_t>(std::max<int64_t>(end, 0), (int64_t)s1 - 1);
        Float sum = 0.0f;
        for (int64_t i = start; i <= end; i++) sum += resize_triangle_kernel(((Float)i - x1) / filter_scale);
        CoeffsLine& l = out[x2]; l.start = (size_t)start;
        for (int64_t i = start; i <= end; i++) l.data.push_back(resize_triangle_kernel(((Float)i - x1) / filter_scale) / sum);
    }
    return out;
}
// resize::resize(w1, h1, w2, h2, RGB f32, Triangle, src, dst): sample_rows (H1 -> H2, tmp stored [x1][y2]) then sample_cols (W1 -> W2)
inline void resize_triangle(size_t w1, size_t h1, size_t w2, size_t h2, const std::vector<Float>& src, std::vector<Float>& dst) {
    std::vector<CoeffsLine> cw = resize_calc_coeffs(w1, w2), ch = resize_calc_coeffs(h1, h2);
    std::vector<Float> tmp(w1 * h2 * 3);
    const size_t stride = w1 * 3;
    for (size_t x1 = 0; x1 < w1; x1++)
        for (size_t y2 = 0; y2 < h2; y2++) {
            Float acc[3] = {0.0f, 0.0f, 0.0f};
            const CoeffsLine& l = ch[y2];
            for (size_t i = 0; i < l.data.size(); i++)
                for (int c = 0; c < 3; c++) acc[c] += src[(l.start + i) * stride + x1 * 3 + c] * l.data[i];
            for (int c = 0; c < 3; c++) tmp[(x1 * h2 + y2) * 3 + c] = acc[c];
        }
    dst.assign(w2 * h2 * 3, 0.0f);
    for (size_t y2 = 0; y2 < h2; y2++)
        for (size_t x2 = 0; x2 < w2; x2++) {
            Float acc[3] = {0.0f, 0.0f, 0.0f};
            const CoeffsLine& l = cw[x2];
            for (size_t i = 0; i < l.data.size(); i++)
                for (int c = 0; c < 3; c++) acc[c] += tmp[((l.start + i) * h2 + y2) * 3 + c] * l.data[i];
            for (int c = 0; c < 3; c++) dst[(y2 * w2 + x2) * 3 + c] = acc[c];
        }
}

inline int log2_usize(size_t n) { int l = 0; while (n >>= 1) l++; return l; }
inline Float lanczos_sinc(Float x, Float tau) {                              // mipmap.rs:38-50
    x = fabsf(x);
    if (x > 1.0f) return 0.0f;
    if (x < 1e-5f) return 1.0f;
    x = x * PI;
    Float s = m_sin(x * tau) / (x * tau);
    Float lanczos = m_sin(x) / x;
    return s * lanczos;
}

struct MIPMap {
    int wrap = FTN_WRAP_REPEAT; int w = 0, h = 0; std::vector<MipLevel> pyramid;
    int levels() const { return (int)pyramid.size(); }

    static MIPMap make(int w, int h, const Float* rgb, int wrap) {          // MIPMap::<Spectrum>::new, mipmap.rs:78-145
        MIPMap m; m.wrap = wrap; m.w = w; m.h = h;
        std::vector<Float> prev(rgb, rgb + (size_t)w * h * 3), cur;
        int n_levels = 1 + log2_usize((size_t)std::max(w, h));
        auto push = [&](const std::vector<Float>& b, int lw, int lh) {
            MipLevel L; L.w = lw; L.h = lh; L.data.resize((size_t)lw * lh);
            for (size_t i = 0; i < L.data.size(); i++) L.data[i] = Spectrum(b[3 * i], b[3 * i + 1], b[3 * i + 2]);
            m.pyramid.push_back(std::move(L));
        };
        push(prev, w, h);
        int cw = w, chh = h;
        for (int l = 1; l < n_levels; l++) {
            int dw = std::max(1, cw / 2), dh = std::max(1, chh / 2);
            resize_triangle((size_t)cw, (size_t)chh, (size_t)dw, (size_t)dh, prev, cur);
            push(cur, dw, dh);
            cw = dw; chh = dh; std::swap(cur, prev);
        }
        return m;
    }
    // MIPMap::new_custom (mipmap.rs:152-243): Lanczos resample to powers of two, 2x2 box pyramid.  Only the reference's own test uses it.
    struct RW { int first; Float w[4]; };
    static std::vector<RW> resample_weights(int old_res, int new_res) {     // :343-367
        std::vector<RW> out((size_t)new_res);
        const Float filter_width = 2.0f;
        for (int i = 0; i < new_res; i++) {
            Float center = ((Float)i + 0.5f) * (Float)old_res / (Float)new_res;
            RW r; r.first = f2i32(floorf((center - filter_width) + 0.5f));
            for (int j = 0; j < 4; j++) { Float pos = (Float)(r.first + j) + 0.5f; r.w[j] = lanczos_sinc((pos - center) / filter_width, 2.0f); }
            Float sum = 0.0f; for (int j = 0; j < 4; j++) sum += r.w[j];      // iter().sum(): left to right from 0.0
            Float inv = 1.0f / sum;
            for (int j = 0; j < 4; j++) r.w[j] *= inv;
            out[(size_t)i] = r;
        }
        return out;
    }
    static int rem_euclid(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }
    static MIPMap make_custom(int w, int h, const std::vector<Spectrum>& image, int wrap) {
        MIPMap m; m.wrap = wrap;
        std::vector<Spectrum> img = image; int rw = w, rh = h;
        auto pow2 = [](int n) { int p = 1; while (p < n) p <<= 1; return p; };
        if ((w & (w - 1)) || (h & (h - 1))) {
            int pw = pow2(w), ph = pow2(h);
            std::vector<RW> sw = resample_weights(w, pw);
            std::vector<Spectrum> res((size_t)pw * ph, Spectrum(0.0f));
            for (int t = 0; t < h; t++)
                for (int s = 0; s < pw; s++)
                    for (int j = 0; j < 4; j++) {
                        int os = sw[(size_t)s].first + j;
                        if (wrap == FTN_WRAP_REPEAT) os = rem_euclid(os, w); else if (wrap == FTN_WRAP_CLAMP) os = std::min(std::max(os, 0), w - 1);
                        if (os >= 0 && os < w) res[(size_t)t * pw + s] += image[(size_t)t * w + os] * sw[(size_t)s].w[j];
                    }
            std::vector<RW> tw = resample_weights(h, ph);
            for (int s = 0; s < pw; s++)
                for (int t = 0; t < ph; t++) {
                    Spectrum wv(0.0f);
                    for (int j = 0; j < 4; j++) {
                        int ot = tw[(size_t)t].first + j;
                        if (wrap == FTN_WRAP_REPEAT) ot = rem_euclid(ot, h); else if (wrap == FTN_WRAP_CLAMP) ot = std::min(std::max(ot, 0), h - 1);
                        if (ot >= 0 && ot < h) wv += res[(size_t)ot * pw + s] * tw[(size_t)t].w[j];
                    }
                    res[(size_t)t * pw + s] = wv;                              // in place, as the reference does (:200)
                }
            img.swap(res); rw = pw; rh = ph;
        }
        m.w = rw; m.h = rh;
        MipLevel L0; L0.w = rw; L0.h = rh; L0.data = img; m.pyramid.push_back(std::move(L0));
        int n_levels = 1 + log2_usize((size_t)std::max(rw, rh));
        int sr = rw, tr = rh;
        for (int l = 1; l < n_levels; l++) {
            sr = std::max(1, sr / 2); tr = std::max(1, tr / 2);
            MipLevel L; L.w = sr; L.h = tr; L.data.resize((size_t)sr * tr);
            const MipLevel& P = m.pyramid.back();
            for (int t = 0; t < tr; t++)
                for (int s = 0; s < sr; s++) {
                    Spectrum sum = texel_of(P, 2 * s, 2 * t, wrap) + texel_of(P, 2 * s + 1, 2 * t, wrap) + texel_of(P, 2 * s, 2 * t + 1, wrap) + texel_of(P, 2 * s + 1, 2 * t + 1, wrap);
                    L.data[(size_t)t * sr + s] = sum * 0.25f;
                }
            m.pyramid.push_back(std::move(L));
        }
        return m;
    }

    static Spectrum texel_of(const MipLevel& L, int s, int t, int wrap) {    // get_texel_from_level :327-341
        if (wrap == FTN_WRAP_REPEAT) { s = rem_euclid(s, L.w); t = rem_euclid(t, L.h); }
        else if (wrap == FTN_WRAP_CLAMP) { s = std::min(std::max(s, 0), L.w - 1); t = std::min(std::max(t, 0), L.h - 1); }
        else if (s < 0 || s >= L.w || t < 0 || t >= L.h) return Spectrum(0.0f);
        return L.data[(size_t)t * L.w + s];
    }
    Spectrum triangle(int level, Vec2 st) const {                           // :294-306
        level = std::min(std::max(level, 0), levels() - 1);
        const MipLevel& L = pyramid[(size_t)level];
        Float s = st.x * (Float)L.w - 0.5f, t = st.y * (Float)L.h - 0.5f;
        int s0 = f2i32(floorf(s)), t0 = f2i32(floorf(t));
        Float ds = s - (Float)s0, dt = t - (Float)t0;
        return texel_of(L, s0, t0, wrap) * (1.0f - ds) * (1.0f - dt) + texel_of(L, s0, t0 + 1, wrap) * (1.0f - ds) * dt +
               texel_of(L, s0 + 1, t0, wrap) * ds * (1.0f - dt) + texel_of(L, s0 + 1, t0 + 1, wrap) * ds * dt;
    }
    Spectrum lookup_trilinear_width(Vec2 st, Float width) const {           // :273-286
        Float level = (Float)levels() - 1.0f + m_log2(fmaxf(width, 1.0e-8f));
        if (level < 0.0f) return triangle(0, st);
        if (level >= (Float)(levels() - 1)) return texel_of(pyramid.back(), 0, 0, wrap);
        int lf = (int)f2usize(floorf(level));
        Float delta = level - truncf(level);                                // f32::fract
        return (1.0f - delta) * triangle(lf, st) + delta * triangle(lf + 1, st);
    }
    Spectrum lookup_trilinear(Vec2 st, Vec2 dst0, Vec2 dst1) const {        // :288-291 (dst0.y without abs, as written)
        Float width = fmaxf(fmaxf(fabsf(dst0.x), dst0.y), fmaxf(fabsf(dst1.x), fabsf(dst1.y)));
        return lookup_trilinear_width(st, 2.0f * width);
    }
};

// ---- Texture::evaluate for the flat texture array of the scene descriptor
struct TextureSet {
    std::vector<ftn_texture> textures; std::vector<MIPMap> images;
    struct TexCoords { Vec2 st, dst_dx, dst_dy; };
    static TexCoords uv_mapping(const ftn_texture& t, const SurfaceInteraction& si) {   // mapping.rs:41-53
        TexCoords c;
        c.dst_dx = Vec2(t.su * si.tex_diffs.dudx, t.sv * si.tex_diffs.dvdx);
        c.dst_dy = Vec2(t.su * si.tex_diffs.dudy, t.sv * si.tex_diffs.dvdy);
        c.st = Vec2(t.su * si.uv.x + t.du, t.sv * si.uv.y + t.dv);
        return c;
    }
    Spectrum evaluate(int id, const SurfaceInteraction& si) const {         // Float textures return their value in every channel
        for (int guard = 0; guard < 64; guard++) {
            const ftn_texture& t = textures[(size_t)id];
            switch (t.kind) {
                case FTN_TEX_CONSTANT: return t.is_float ? Spectrum(t.value[0]) : Spectrum(t.value[0], t.value[1], t.value[2]);
                case FTN_TEX_UV: {                                          // uv.rs:17-23
                    TexCoords c = uv_mapping(t, si);
                    return Spectrum(c.st.x - floorf(c.st.x), c.st.y - floorf(c.st.y), 0.0f);
                }
                case FTN_TEX_CHECKERBOARD: {                                // checkerboard.rs:49-64, AAMethod::None
                    TexCoords c = uv_mapping(t, si);
                    int32_t a = f2i32(floorf(c.st.x)), b = f2i32(floorf(c.st.y));
                    id = ((int32_t)((uint32_t)a + (uint32_t)b) % 2 == 0) ? t.tex1 : t.tex2;   // i32 add (wrapping in release), Rust % keeps the sign
                    continue;
                }
                case FTN_TEX_IMAGE: {                                       // image.rs:28-34
                    TexCoords c = uv_mapping(t, si);
                    return images[(size_t)t.image].lookup_trilinear(c.st, c.dst_dx, c.dst_dy);
                }
                default: return Spectrum(0.0f);
            }
        }
        return Spectrum(0.0f);
    }
    // a material whose parameters are textures, evaluated at `si`: the ftn_material the constant-parameter code expects
    ftn_material resolve(const ftn_material& m, const ftn_material_textures* mt, const SurfaceInteraction& si) const {
        if (!mt) return m;
        ftn_material r = m;
        if (mt->a >= 0) { Spectrum v = evaluate(mt->a, si); r.a[0] = v[0]; r.a[1] = v[1]; r.a[2] = v[2]; }
        if (mt->b >= 0) { Spectrum v = evaluate(mt->b, si); r.b[0] = v[0]; r.b[1] = v[1]; r.b[2] = v[2]; }
        if (mt->s0 >= 0) r.s0 = evaluate(mt->s0, si)[0];
        if (mt->s1 >= 0) r.s1 = evaluate(mt->s1, si)[0];
        if (mt->s2 >= 0) r.s2 = evaluate(mt->s2, si)[0];
        return r;
    }
};

inline int build_textures(const ftn_scene_desc* d, TextureSet* ts, std::vector<ftn_material_textures>* mtex) {
    ts->textures.assign(d->textures, d->textures + (d->textures ? d->n_textures : 0));
    ts->images.clear();
    for (uint32_t i = 0; i < (d->images ? d->n_images : 0); i++) {
        const ftn_image& im = d->images[i];
        if (im.width == 0 || im.height == 0 || !im.texels || im.wrap > FTN_WRAP_CLAMP) return FTN_ERR_INVALID_ARGUMENT;
        ts->images.push_back(MIPMap::make((int)im.width, (int)im.height, im.texels, (int)im.wrap));
    }
    const int nt = (int)ts->textures.size(), ni = (int)ts->images.size();
    for (const ftn_texture& t : ts->textures) {
        if (t.kind > FTN_TEX_IMAGE) return FTN_ERR_INVALID_ARGUMENT;
        if (t.kind == FTN_TEX_CHECKERBOARD && (t.tex1 < 0 || t.tex1 >= nt || t.tex2 < 0 || t.tex2 >= nt)) return FTN_ERR_INVALID_ARGUMENT;
        if (t.kind == FTN_TEX_IMAGE && (t.image < 0 || t.image >= ni)) return FTN_ERR_INVALID_ARGUMENT;
    }
    mtex->clear();
    if (d->material_textures) {
        mtex->assign(d->material_textures, d->material_textures + d->n_materials);
        for (const ftn_material_textures& m : *mtex)
            for (int32_t id : {m.a, m.b, m.s0, m.s1, m.s2}) if (id >= nt) return FTN_ERR_INVALID_ARGUMENT;
    }
    return FTN_OK;
}

}  // namespace orc
