/*
 * ftn_pbrt.cpp -- host-side scene ingestion: the subset of the PBRT v3 scene format that fountain's loader evaluates
 * (src/loaders/pbrt.rs:178-605, defaults from src/loaders/constructors.rs:38-359) and an ASCII / binary-little-endian PLY
 * reader for `Shape "plymesh"` (constructors.rs:94-190), producing the flat descriptors of include/fountain_hip.h.
 * Every number goes through the same ftn_* constructors the rest of the library uses, so a parsed file and a scene assembled
 * call by call are bit-identical.  Reference quirks kept: `ReverseOrientation` sets (does not toggle) the flag (pbrt.rs:203-205);
 * plastic reads "ks" in lower case (constructors.rs:232); point/distant lights ignore the CTM; Integrator / PixelFilter /
 * Accelerator statements are ignored (pbrt.rs:528-530); ObjectBegin/End are unimplemented in the reference -> FTN_ERR_UNSUPPORTED.
 * Texture statements: checkerboard (spectrum / float), uv and imagemap (OpenEXR files) as in pbrt.rs:362-385.
 */
#include "../../include/fountain_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct Param { std::string type, name; std::vector<float> f; std::vector<int> i; std::vector<std::string> s; std::vector<int> b; };
typedef std::map<std::string, Param> ParamSet;

struct Tok { enum Kind { END, WORD, STR, NUM, LB, RB } kind; std::string text; double num; };

struct Lexer {
    std::string src; size_t p = 0;
    Tok next() {
        for (;;) {
            while (p < src.size() && isspace((unsigned char)src[p])) p++;
            if (p < src.size() && src[p] == '#') { while (p < src.size() && src[p] != '\n') p++; continue; }
            break;
        }
        Tok t; t.num = 0;
        if (p >= src.size()) { t.kind = Tok::END; return t; }
        char c = src[p];
        if (c == '[') { p++; t.kind = Tok::LB; return t; }
        if (c == ']') { p++; t.kind = Tok::RB; return t; }
        if (c == '"') { size_t e = src.find('"', p + 1); if (e == std::string::npos) e = src.size(); t.kind = Tok::STR; t.text = src.substr(p + 1, e - p - 1); p = e + 1; return t; }
        size_t s = p;
        while (p < src.size() && !isspace((unsigned char)src[p]) && src[p] != '[' && src[p] != ']' && src[p] != '"' && src[p] != '#') p++;
        t.text = src.substr(s, p - s);
        char* endp = nullptr; double v = strtod(t.text.c_str(), &endp);
        if (endp && *endp == 0 && !t.text.empty() && (isdigit((unsigned char)t.text[0]) || t.text[0] == '-' || t.text[0] == '+' || t.text[0] == '.')) { t.kind = Tok::NUM; t.num = v; }
        else t.kind = Tok::WORD;
        return t;
    }
};

struct Mesh { std::vector<float> P, N, UV, S; std::vector<uint32_t> idx; };

struct GState { int material; int area; bool rev; };

}  // namespace

struct ftn_pbrt {
    std::string error, base_dir, film_name = "render.exr";
    /* header */
    ftn_transform header_tf, camera_tf; ParamSet camera_params, sampler_params, film_params; bool has_camera = false;
    ftn_camera_desc camera; ftn_film_desc film; int spp = 16;
    /* world */
    std::vector<GState> gs; std::vector<ftn_transform> tf;
    std::map<std::string, int> named_materials;
    std::vector<ftn_prim> prims; std::vector<uint32_t> tri_indices, tri_mesh; std::vector<float> P, N, UV, T; bool any_n = false, any_uv = false, any_t = false;
    std::vector<ftn_mesh> meshes; std::vector<ftn_sphere> spheres; std::vector<ftn_material> materials; std::vector<float> area_emit;
    std::vector<ftn_texture> textures; std::vector<ftn_material_textures> mtex; std::map<std::string, int> spectrum_textures, float_textures;
    struct Img { uint32_t w, h, wrap; std::vector<float> texels; }; std::vector<Img> image_store; std::vector<ftn_image> images;
    std::vector<ftn_light> lights; std::vector<ftn_envmap> envmaps; struct Env { uint32_t w, h; std::vector<float> texels; }; std::vector<Env> env_store;
    ftn_scene_desc desc;
};

namespace {

int fail(ftn_pbrt* S, int code, const std::string& m) { S->error = m; return code; }

bool parse_params(Lexer& L, Tok& t, ParamSet* out, std::string* err) {
    /* "type name" value | [values]  ... until the next directive word */
    while (t.kind == Tok::STR) {
        std::istringstream ss(t.text); Param p; ss >> p.type >> p.name;
        if (p.name.empty()) { *err = "malformed parameter declaration '" + t.text + "'"; return false; }
        t = L.next();
        std::vector<Tok> vals;
        if (t.kind == Tok::LB) { for (t = L.next(); t.kind != Tok::RB && t.kind != Tok::END; t = L.next()) vals.push_back(t); t = L.next(); }
        else { vals.push_back(t); t = L.next(); }
        for (const Tok& v : vals) {
            if (p.type == "integer") p.i.push_back((int)v.num);
            else if (p.type == "bool") p.b.push_back(v.text == "true" ? 1 : 0);
            else if (p.type == "string" || p.type == "texture") p.s.push_back(v.text);
            else p.f.push_back((float)v.num);
        }
        (*out)[p.name] = p;
    }
    return true;
}
float getf(const ParamSet& ps, const char* n, float def) { auto it = ps.find(n); return (it != ps.end() && it->second.type == "float" && !it->second.f.empty()) ? it->second.f[0] : def; }
int geti(const ParamSet& ps, const char* n, int def) { auto it = ps.find(n); return (it != ps.end() && it->second.type == "integer" && !it->second.i.empty()) ? it->second.i[0] : def; }
bool getb(const ParamSet& ps, const char* n, bool def) { auto it = ps.find(n); return (it != ps.end() && it->second.type == "bool" && !it->second.b.empty()) ? it->second.b[0] != 0 : def; }
bool is_spectrum(const Param& p) { return p.type == "rgb" || p.type == "color"; }
/* lookup_texture (pbrt.rs:142-147): spectrum textures first, then float textures */
int lookup_texture(ftn_pbrt* S, const std::string& name, int* idx, bool* is_float) {
    auto a = S->spectrum_textures.find(name);
    if (a != S->spectrum_textures.end()) { *idx = a->second; *is_float = false; return FTN_OK; }
    auto b = S->float_textures.find(name);
    if (b != S->float_textures.end()) { *idx = b->second; *is_float = true; return FTN_OK; }
    return fail(S, FTN_ERR_INVALID_ARGUMENT, "TextureError: " + name);
}
/* make_param_set resolves every "texture" parameter up front (pbrt.rs:114-131): an unknown name fails the statement */
int check_texture_params(ftn_pbrt* S, const ParamSet& ps) {
    for (const auto& kv : ps) if (kv.second.type == "texture") { int i; bool f; if (kv.second.s.empty()) return fail(S, FTN_ERR_INVALID_ARGUMENT, "empty texture parameter"); int rc = lookup_texture(S, kv.second.s[0], &i, &f); if (rc) return rc; }
    return FTN_OK;
}
/* get_texture_or_default::<Spectrum> (loaders/mod.rs:227-236): an rgb constant, a spectrum texture, else the default (a float
 * texture fails the conversion and falls back to the default as well) */
int get_rgb(ftn_pbrt* S, const ParamSet& ps, const char* n, const float def[3], float out[3], int32_t* tex) {
    auto it = ps.find(n);
    out[0] = def[0]; out[1] = def[1]; out[2] = def[2]; *tex = -1;
    if (it == ps.end()) return FTN_OK;
    if (it->second.type == "texture") { int i; bool f; int rc = lookup_texture(S, it->second.s[0], &i, &f); if (rc) return rc; if (!f) *tex = i; return FTN_OK; }
    if (is_spectrum(it->second) && it->second.f.size() >= 3) { out[0] = it->second.f[0]; out[1] = it->second.f[1]; out[2] = it->second.f[2]; }
    return FTN_OK;
}
int get_ftex(ftn_pbrt* S, const ParamSet& ps, const char* n, float def, float* out, int32_t* tex) {
    auto it = ps.find(n); *out = def; *tex = -1;
    if (it == ps.end()) return FTN_OK;
    if (it->second.type == "texture") { int i; bool f; int rc = lookup_texture(S, it->second.s[0], &i, &f); if (rc) return rc; if (f) *tex = i; return FTN_OK; }
    if (it->second.type == "float" && !it->second.f.empty()) *out = it->second.f[0];
    return FTN_OK;
}
int new_texture(ftn_pbrt* S, uint32_t kind, bool is_float, const float value[3], int tex1, int tex2, int image, const float map[4]) {
    ftn_texture t; memset(&t, 0, sizeof(t));
    t.kind = kind; t.is_float = is_float ? 1u : 0u; t.tex1 = tex1; t.tex2 = tex2; t.image = image;
    if (value) { t.value[0] = value[0]; t.value[1] = value[1]; t.value[2] = value[2]; }
    t.su = map ? map[0] : 1.0f; t.sv = map ? map[1] : 1.0f; t.du = map ? map[2] : 0.0f; t.dv = map ? map[3] : 0.0f;
    S->textures.push_back(t);
    return (int)S->textures.size() - 1;
}
/* get_texture_or_const (loaders/mod.rs:213-225): required; a texture of the other output type is an error */
int texture_or_const(ftn_pbrt* S, const ParamSet& ps, const char* n, bool is_float, int* out) {
    auto it = ps.find(n);
    if (it == ps.end()) return fail(S, FTN_ERR_INVALID_ARGUMENT, std::string("ParamError: missing ") + n);
    if (it->second.type == "texture") {
        int i; bool f; int rc = lookup_texture(S, it->second.s[0], &i, &f); if (rc) return rc;
        if (f != is_float) return fail(S, FTN_ERR_INVALID_ARGUMENT, std::string("ParamError: texture for ") + n + " has the wrong output type");
        *out = i; return FTN_OK;
    }
    float v[3];
    if (is_float) { if (it->second.type != "float" || it->second.f.empty()) return fail(S, FTN_ERR_INVALID_ARGUMENT, std::string("ParamError: ") + n); v[0] = v[1] = v[2] = it->second.f[0]; }
    else { if (!is_spectrum(it->second) || it->second.f.size() < 3) return fail(S, FTN_ERR_INVALID_ARGUMENT, std::string("ParamError: ") + n); v[0] = it->second.f[0]; v[1] = it->second.f[1]; v[2] = it->second.f[2]; }
    *out = new_texture(S, FTN_TEX_CONSTANT, is_float, v, -1, -1, -1, nullptr);
    return FTN_OK;
}

/* ---- PLY (constructors.rs:94-190): x y z [nx ny nz] [u v] float vertices, uchar-counted int/uint triangle faces */
int load_ply(ftn_pbrt* S, const std::string& path, Mesh* m) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return fail(S, FTN_ERR_INVALID_ARGUMENT, "cannot open " + path);
    std::string line; std::getline(f, line);
    if (line.substr(0, 3) != "ply") return fail(S, FTN_ERR_INVALID_ARGUMENT, path + ": not a PLY file");
    bool binary = false; size_t nv = 0, nf = 0; std::string cur; std::vector<std::string> vprops, vtypes; std::string cnt_t = "uchar", idx_t = "int";
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ss(line); std::string w; ss >> w;
        if (w == "format") { ss >> w; if (w == "binary_little_endian") binary = true; else if (w != "ascii") return fail(S, FTN_ERR_UNSUPPORTED, "PLY format " + w); }
        else if (w == "element") { ss >> cur; if (cur == "vertex") ss >> nv; else if (cur == "face") ss >> nf; }
        else if (w == "property") {
            std::string ty; ss >> ty;
            if (cur == "vertex") { std::string name; ss >> name; vtypes.push_back(ty); vprops.push_back(name); }
            else if (cur == "face" && ty == "list") { std::string name; ss >> cnt_t >> idx_t >> name; }
        } else if (w == "end_header") break;
    }
    auto col = [&](const char* n) { for (size_t i = 0; i < vprops.size(); i++) if (vprops[i] == n) return (int)i; return -1; };
    const int cx = col("x"), cy = col("y"), cz = col("z"), cnx = col("nx"), cny = col("ny"), cnz = col("nz"), cu = col("u"), cv = col("v");
    if (cx < 0 || cy < 0 || cz < 0) return fail(S, FTN_ERR_INVALID_ARGUMENT, "Ply file is missing vertex coordinates");
    for (const std::string& t : vtypes) if (t != "float" && t != "float32") return fail(S, FTN_ERR_UNSUPPORTED, "PLY vertex property type " + t);
    const bool hn = cnx >= 0 && cny >= 0 && cnz >= 0, huv = cu >= 0 && cv >= 0;
    m->P.resize(3 * nv); if (hn) m->N.resize(3 * nv); if (huv) m->UV.resize(2 * nv);
    std::vector<float> row(vprops.size());
    for (size_t i = 0; i < nv; i++) {
        if (binary) f.read((char*)row.data(), row.size() * 4); else for (float& x : row) f >> x;
        m->P[3 * i] = row[cx]; m->P[3 * i + 1] = row[cy]; m->P[3 * i + 2] = row[cz];
        if (hn) { m->N[3 * i] = row[cnx]; m->N[3 * i + 1] = row[cny]; m->N[3 * i + 2] = row[cnz]; }
        if (huv) { m->UV[2 * i] = row[cu]; m->UV[2 * i + 1] = row[cv]; }
    }
    m->idx.reserve(3 * nf);
    for (size_t i = 0; i < nf; i++) {
        int n = 0; int v[3] = {0, 0, 0};
        if (binary) {
            if (cnt_t == "uchar" || cnt_t == "uint8") { unsigned char c; f.read((char*)&c, 1); n = c; } else { int c; f.read((char*)&c, 4); n = c; }
            if (n != 3) return fail(S, FTN_ERR_UNSUPPORTED, "Face with unsupported vertex count found");
            f.read((char*)v, 12);
        } else { f >> n; if (n != 3) return fail(S, FTN_ERR_UNSUPPORTED, "Face with unsupported vertex count found"); f >> v[0] >> v[1] >> v[2]; }
        for (int k = 0; k < 3; k++) m->idx.push_back((uint32_t)v[k]);
    }
    if (!f && !f.eof()) return fail(S, FTN_ERR_INVALID_ARGUMENT, path + ": truncated PLY");
    return FTN_OK;
}

int add_material(ftn_pbrt* S, const std::string& name, const ParamSet& ps, int* out) {
    ftn_material m; memset(&m, 0, sizeof(m)); m.remap_roughness = 1;
    ftn_material_textures mt; mt.a = mt.b = mt.s0 = mt.s1 = mt.s2 = -1; mt._pad[0] = mt._pad[1] = mt._pad[2] = 0;
    int rc;
    const float d05[3] = {0.5f, 0.5f, 0.5f}, d1[3] = {1, 1, 1}, d09[3] = {0.9f, 0.9f, 0.9f}, d025[3] = {0.25f, 0.25f, 0.25f};
    if (name == "matte") { m.type = FTN_MAT_MATTE; if ((rc = get_rgb(S, ps, "Kd", d05, m.a, &mt.a)) || (rc = get_ftex(S, ps, "sigma", 0.0f, &m.s0, &mt.s0))) return rc; }
    else if (name == "glass") {
        m.type = FTN_MAT_GLASS;
        if ((rc = get_rgb(S, ps, "Kr", d1, m.a, &mt.a)) || (rc = get_rgb(S, ps, "Kt", d1, m.b, &mt.b)) || (rc = get_ftex(S, ps, "uroughness", 0.0f, &m.s1, &mt.s1)) ||
            (rc = get_ftex(S, ps, "vroughness", 0.0f, &m.s2, &mt.s2)) || (rc = get_ftex(S, ps, "eta", 1.5f, &m.s0, &mt.s0))) return rc;
        m.remap_roughness = getb(ps, "remaproughness", true) ? 1 : 0;
    } else if (name == "mirror") { m.type = FTN_MAT_MIRROR; if ((rc = get_rgb(S, ps, "Kr", d09, m.a, &mt.a))) return rc; }
    else if (name == "metal") {                                 /* make_metal_material constructors.rs:213-230 */
        m.type = FTN_MAT_METAL;
        for (int which = 0; which < 2; which++) {               /* get_texture_or_const: required, spectrum-valued */
            const char* n = which ? "k" : "eta"; float* dst = which ? m.b : m.a; int32_t* slot = which ? &mt.b : &mt.a;
            auto it = ps.find(n);
            if (it == ps.end()) return fail(S, FTN_ERR_INVALID_ARGUMENT, "metal needs eta and k (constructors.rs:215-216)");
            if (it->second.type == "texture") {
                int i; bool f; if ((rc = lookup_texture(S, it->second.s[0], &i, &f))) return rc;
                if (f) return fail(S, FTN_ERR_INVALID_ARGUMENT, std::string("ParamError: texture for ") + n + " has the wrong output type");
                *slot = i;
            } else if (is_spectrum(it->second) && it->second.f.size() >= 3) { dst[0] = it->second.f[0]; dst[1] = it->second.f[1]; dst[2] = it->second.f[2]; }
            else return fail(S, FTN_ERR_INVALID_ARGUMENT, std::string("ParamError: ") + n);
        }
        float rough; int32_t trough; if ((rc = get_ftex(S, ps, "roughness", 0.01f, &rough, &trough))) return rc;
        auto present = [&](const char* n) { auto it = ps.find(n); if (it == ps.end()) return false; if (it->second.type == "float") return true;
                                            if (it->second.type == "texture") { int i; bool f; return lookup_texture(S, it->second.s[0], &i, &f) == FTN_OK && f; } return false; };
        if (present("uroughness") && present("vroughness")) { if ((rc = get_ftex(S, ps, "uroughness", 0.0f, &m.s1, &mt.s1)) || (rc = get_ftex(S, ps, "vroughness", 0.0f, &m.s2, &mt.s2))) return rc; }
        else { m.s1 = rough; m.s2 = rough; mt.s1 = trough; mt.s2 = trough; }
        m.remap_roughness = getb(ps, "remaproughness", true) ? 1 : 0;
    } else if (name == "plastic") {
        m.type = FTN_MAT_PLASTIC;
        if ((rc = get_rgb(S, ps, "Kd", d025, m.a, &mt.a)) || (rc = get_rgb(S, ps, "ks", d025, m.b, &mt.b)) || (rc = get_ftex(S, ps, "roughness", 0.1f, &m.s1, &mt.s1))) return rc;
        m.remap_roughness = getb(ps, "remaproughness", true) ? 1 : 0;
    } else return fail(S, FTN_ERR_INVALID_ARGUMENT, "UnknownName(" + name + ")");
    S->materials.push_back(m); S->mtex.push_back(mt);
    *out = (int)S->materials.size() - 1;
    return FTN_OK;
}

/* PbrtSceneBuilder::texture (pbrt.rs:362-385) + make_checkerboard_* / make_uv_spect / make_imagemap_spect (constructors.rs:247-318) */
int add_texture(ftn_pbrt* S, const std::string& name, std::string ty, const std::string& cls, const ParamSet& ps) {
    if (ty == "color") ty = "spectrum";
    const bool is_float = ty == "float";
    if (!((ty == "spectrum" && (cls == "checkerboard" || cls == "uv" || cls == "imagemap")) || (is_float && cls == "checkerboard")))
        return fail(S, FTN_ERR_INVALID_ARGUMENT, "UnknownName(" + ty + " " + cls + ")");
    { auto it = ps.find("mapping"); if (it != ps.end() && !it->second.s.empty() && it->second.s[0] != "uv") return fail(S, FTN_ERR_INVALID_ARGUMENT, "Unknown mapping type " + it->second.s[0]); }
    const float map[4] = {getf(ps, "uscale", 1.0f), getf(ps, "vscale", 1.0f), getf(ps, "udelta", 0.0f), getf(ps, "vdelta", 0.0f)};
    int idx, rc;
    if (cls == "checkerboard") {
        int t1, t2;
        if ((rc = texture_or_const(S, ps, "tex1", is_float, &t1)) || (rc = texture_or_const(S, ps, "tex2", is_float, &t2))) return rc;
        idx = new_texture(S, FTN_TEX_CHECKERBOARD, is_float, nullptr, t1, t2, -1, map);
    } else if (cls == "uv") idx = new_texture(S, FTN_TEX_UV, false, nullptr, -1, -1, -1, map);
    else {
        auto fn = ps.find("filename");
        if (fn == ps.end() || fn->second.s.empty()) return fail(S, FTN_ERR_INVALID_ARGUMENT, "ParamError: filename");
        const std::string file = S->base_dir + "/" + fn->second.s[0];
        if (file.size() < 4 || file.substr(file.size() - 4) != ".exr") return fail(S, FTN_ERR_UNSUPPORTED, "image maps are read from OpenEXR files only");
        uint32_t wrap = FTN_WRAP_REPEAT;
        { auto it = ps.find("wrap"); if (it != ps.end() && !it->second.s.empty()) { const std::string& w = it->second.s[0];
            if (w == "repeat") wrap = FTN_WRAP_REPEAT; else if (w == "black") wrap = FTN_WRAP_BLACK; else if (w == "clamp") wrap = FTN_WRAP_CLAMP; else return fail(S, FTN_ERR_INVALID_ARGUMENT, "Unknown repeat type " + w); } }
        ftn_pbrt::Img im; im.wrap = wrap;
        if ((rc = ftn_exr_read(file.c_str(), &im.w, &im.h, nullptr))) return fail(S, rc, ftn_imageio_last_error());
        im.texels.resize((size_t)im.w * im.h * 3);
        if ((rc = ftn_exr_read(file.c_str(), &im.w, &im.h, im.texels.data()))) return fail(S, rc, ftn_imageio_last_error());
        /* load_mipmap (imageio/mod.rs:86-117): `gamma` given -> as given, otherwise false for .exr (the only extension read here); the
         * gamma step first, then texel * scale, then the y flip */
        if (getb(ps, "gamma", false)) (void)ftn_image_inverse_gamma(im.texels.data(), im.texels.size());
        const float scale = getf(ps, "scale", 1.0f);
        for (float& t : im.texels) t = t * scale;
        for (uint32_t y = 0; y < im.h / 2; y++) for (uint32_t x = 0; x < im.w * 3; x++) std::swap(im.texels[(size_t)y * im.w * 3 + x], im.texels[(size_t)(im.h - 1 - y) * im.w * 3 + x]);
        S->image_store.push_back(std::move(im));
        idx = new_texture(S, FTN_TEX_IMAGE, false, nullptr, -1, -1, (int)S->image_store.size() - 1, map);
    }
    (is_float ? S->float_textures : S->spectrum_textures)[name] = idx;
    return FTN_OK;
}

int add_mesh(ftn_pbrt* S, const Mesh& m) {      /* TriangleMesh::new (triangle.rs:29-74) + one GeometricPrimitive per triangle */
    const ftn_transform& tf = S->tf.back(); const GState& g = S->gs.back();
    const size_t nv = m.P.size() / 3, base = S->P.size() / 3;
    ftn_mesh fm; fm.has_normals = m.N.empty() ? 0 : 1; fm.has_uvs = m.UV.empty() ? 0 : 1; fm.reverse_orientation = g.rev ? 1 : 0; fm.has_tangents = m.S.empty() ? 0 : 1;
    fm.flip_normals = (g.rev != (ftn_transform_swaps_handedness(&tf) != 0)) ? 1 : 0;
    S->meshes.push_back(fm);
    S->P.resize(3 * (base + nv)); S->N.resize(3 * (base + nv), 0.0f); S->UV.resize(2 * (base + nv), 0.0f); S->T.resize(3 * (base + nv), 0.0f);
    if (!m.S.empty()) { for (size_t v = 0; v < nv; v++) ftn_transform_vector(&tf, m.S.data() + 3 * v, S->T.data() + 3 * (base + v)); S->any_t = true; }      /* triangle.rs:53-58 */
    ftn_transform_points(&tf, nv, m.P.data(), S->P.data() + 3 * base);
    if (!m.N.empty()) { ftn_transform_normals(&tf, nv, m.N.data(), S->N.data() + 3 * base); S->any_n = true; }
    if (!m.UV.empty()) { memcpy(S->UV.data() + 2 * base, m.UV.data(), m.UV.size() * 4); S->any_uv = true; }
    const uint32_t first_tri = (uint32_t)(S->tri_indices.size() / 3), mesh_id = (uint32_t)S->meshes.size() - 1;
    for (uint32_t v : m.idx) { if (v >= nv) return fail(S, FTN_ERR_INVALID_ARGUMENT, "vertex index out of range"); S->tri_indices.push_back(v + (uint32_t)base); }
    for (size_t t = 0; t < m.idx.size() / 3; t++) {
        S->tri_mesh.push_back(mesh_id);
        ftn_prim p; p.shape_kind = FTN_SHAPE_TRIANGLE; p.shape_index = first_tri + (uint32_t)t; p.material = g.material; p.area_emit = g.area;
        S->prims.push_back(p);
    }
    return FTN_OK;
}

int do_transform(ftn_pbrt* S, const std::string& w, Lexer& L, Tok& t, ftn_transform* ctm) {       /* eval_transform_stmt pbrt.rs:567-603 */
    auto nums = [&](size_t n, float* out) { bool br = false; if (t.kind == Tok::LB) { br = true; t = L.next(); } for (size_t i = 0; i < n; i++) { if (t.kind != Tok::NUM) return false; out[i] = (float)t.num; t = L.next(); } if (br) { if (t.kind != Tok::RB) return false; t = L.next(); } return true; };
    ftn_transform x; float v[16]; int rc = FTN_OK;
    if (w == "Identity") { ftn_transform_identity(ctm); return FTN_OK; }
    if (w == "Translate") { if (!nums(3, v)) return fail(S, FTN_ERR_INVALID_ARGUMENT, "Translate needs 3 numbers"); ftn_transform_translate(v, &x); }
    else if (w == "Scale") { if (!nums(3, v)) return fail(S, FTN_ERR_INVALID_ARGUMENT, "Scale needs 3 numbers"); ftn_transform_scale(v[0], v[1], v[2], &x); }
    else if (w == "Rotate") { if (!nums(4, v)) return fail(S, FTN_ERR_INVALID_ARGUMENT, "Rotate needs 4 numbers"); rc = ftn_transform_rotate(v[0], v + 1, &x); }
    else if (w == "LookAt") { if (!nums(9, v)) return fail(S, FTN_ERR_INVALID_ARGUMENT, "LookAt needs 9 numbers"); rc = ftn_transform_look_at(v, v + 3, v + 6, &x); }
    else if (w == "Transform") { if (!nums(16, v)) return fail(S, FTN_ERR_INVALID_ARGUMENT, "Transform needs 16 numbers"); rc = ftn_transform_from_flat(v, ctm); if (rc) return fail(S, rc, ftn_last_error()); return FTN_OK; }
    else if (w == "ConcatTransform") { if (!nums(16, v)) return fail(S, FTN_ERR_INVALID_ARGUMENT, "ConcatTransform needs 16 numbers"); rc = ftn_transform_from_flat(v, &x); }
    else return fail(S, FTN_ERR_UNSUPPORTED, w + " is unimplemented in the reference (pbrt.rs:592-597)");
    if (rc) return fail(S, rc, ftn_last_error());
    ftn_transform r; ftn_transform_mul(ctm, &x, &r); *ctm = r;         /* *current_tf * stmt */
    return FTN_OK;
}

bool is_transform_word(const std::string& w) { return w == "Identity" || w == "Translate" || w == "Scale" || w == "Rotate" || w == "LookAt" || w == "Transform" || w == "ConcatTransform" || w == "CoordinateSystem" || w == "CoordSysTransform"; }

int read_file(const std::string& path, std::string* out) {
    std::ifstream f(path, std::ios::binary); if (!f) return -1;
    std::stringstream ss; ss << f.rdbuf(); *out = ss.str(); return 0;
}

int parse(ftn_pbrt* S, const std::string& path) {
    std::string src;
    if (read_file(path, &src)) return fail(S, FTN_ERR_INVALID_ARGUMENT, "cannot open " + path);
    /* parse_with_includes: textual Include before evaluation */
    for (size_t pos; (pos = src.find("Include")) != std::string::npos;) {
        size_t q0 = src.find('"', pos), q1 = q0 == std::string::npos ? q0 : src.find('"', q0 + 1);
        if (q1 == std::string::npos) break;
        std::string inc, name = src.substr(q0 + 1, q1 - q0 - 1);
        if (read_file(S->base_dir + "/" + name, &inc)) return fail(S, FTN_ERR_INVALID_ARGUMENT, "cannot open include " + name);
        src = src.substr(0, pos) + inc + src.substr(q1 + 1);
    }
    Lexer L; L.src = src;
    Tok t = L.next();
    bool world = false;
    ftn_transform_identity(&S->header_tf); ftn_transform_identity(&S->camera_tf);
    const float d05[3] = {0.5f, 0.5f, 0.5f};
    int rc;
    while (t.kind != Tok::END) {
        if (t.kind != Tok::WORD) return fail(S, FTN_ERR_INVALID_ARGUMENT, "expected a directive, got '" + t.text + "'");
        const std::string w = t.text; t = L.next();
        if (is_transform_word(w)) { if ((rc = do_transform(S, w, L, t, world ? &S->tf.back() : &S->header_tf))) return rc; continue; }
        if (w == "WorldBegin") {
            world = true;
            S->tf.assign(1, ftn_transform()); ftn_transform_identity(&S->tf[0]);
            ParamSet none; int mat; if ((rc = add_material(S, "matte", none, &mat))) return rc;      /* default material (pbrt.rs:88-96) */
            S->gs.assign(1, GState{mat, -1, false});
            continue;
        }
        if (w == "WorldEnd") break;
        if (w == "AttributeBegin") { S->gs.push_back(S->gs.back()); S->tf.push_back(S->tf.back()); continue; }
        if (w == "AttributeEnd") { if (S->gs.size() < 2) return fail(S, FTN_ERR_INVALID_ARGUMENT, "unbalanced AttributeEnd"); S->gs.pop_back(); S->tf.pop_back(); continue; }
        if (w == "TransformBegin") { S->tf.push_back(S->tf.back()); continue; }
        if (w == "TransformEnd") { if (S->tf.size() < 2) return fail(S, FTN_ERR_INVALID_ARGUMENT, "unbalanced TransformEnd"); S->tf.pop_back(); continue; }
        if (w == "ReverseOrientation") { S->gs.back().rev = true; continue; }
        if (w == "ObjectBegin" || w == "ObjectEnd" || w == "MakeNamedMedium" || w == "MediumInterface")
            return fail(S, FTN_ERR_UNSUPPORTED, w + " is unimplemented!() in the reference (pbrt.rs:196-201, :242-247)");
        if (w == "ObjectInstance") { t = L.next(); continue; }                                       /* silently dropped (pbrt.rs:214) */
        if (w == "NamedMaterial") {
            auto it = S->named_materials.find(t.text);
            if (it == S->named_materials.end()) return fail(S, FTN_ERR_INVALID_ARGUMENT, "MaterialError: " + t.text);
            S->gs.back().material = it->second; t = L.next(); continue;
        }
        /* directives of the form  Name "type" params... */
        std::string type;
        if (w == "Texture") {   /* Texture "name" "type" "class" params */
            std::string names[3];
            for (int k = 0; k < 3; k++) { if (t.kind != Tok::STR) return fail(S, FTN_ERR_INVALID_ARGUMENT, "Texture: expected name, type and class"); names[k] = t.text; t = L.next(); }
            ParamSet tps; std::string terr;
            if (!parse_params(L, t, &tps, &terr)) return fail(S, FTN_ERR_INVALID_ARGUMENT, terr);
            if ((rc = check_texture_params(S, tps)) || (rc = add_texture(S, names[0], names[1], names[2], tps))) return rc;
            continue;
        }
        if (t.kind != Tok::STR) return fail(S, FTN_ERR_INVALID_ARGUMENT, w + ": expected a quoted name");
        type = t.text; t = L.next();
        ParamSet ps; std::string perr;
        if (!parse_params(L, t, &ps, &perr)) return fail(S, FTN_ERR_INVALID_ARGUMENT, perr);
        if (world && (rc = check_texture_params(S, ps))) return rc;
        if (w == "Camera") { S->camera_params = ps; S->camera_params["name"].s = {type}; S->camera_tf = S->header_tf; S->has_camera = true; }
        else if (w == "Sampler") S->sampler_params = ps;
        else if (w == "Film") S->film_params = ps;
        else if (w == "PixelFilter" || w == "Integrator" || w == "Accelerator") { /* ignored: pbrt.rs:528-530 */ }
        else if (w == "Material") { int m; if ((rc = add_material(S, type, ps, &m))) return rc; S->gs.back().material = m; }
        else if (w == "MakeNamedMaterial") {
            auto it = ps.find("type"); if (it == ps.end() || it->second.s.empty()) return fail(S, FTN_ERR_INVALID_ARGUMENT, "MakeNamedMaterial needs a string type");
            int m; if ((rc = add_material(S, it->second.s[0], ps, &m))) return rc; S->named_materials[type] = m;
        } else if (w == "AreaLightSource") {
            if (type != "diffuse") return fail(S, FTN_ERR_INVALID_ARGUMENT, "UnknownName(" + type + ")");
            const float one[3] = {1, 1, 1}; float Lr[3]; int32_t nt; if ((rc = get_rgb(S, ps, "L", one, Lr, &nt))) return rc;
            S->area_emit.insert(S->area_emit.end(), Lr, Lr + 3); S->gs.back().area = (int)S->area_emit.size() / 3 - 1;
        } else if (w == "LightSource") {
            ftn_light l; memset(&l, 0, sizeof(l)); l.envmap = -1;
            const float one[3] = {1, 1, 1}, zero[3] = {0, 0, 0}, zup[3] = {0, 0, 1};
            auto vec3 = [&](const char* n, const float def[3], float out[3]) { auto it = ps.find(n); for (int k = 0; k < 3; k++) out[k] = (it != ps.end() && it->second.f.size() >= 3) ? it->second.f[k] : def[k]; };
            float scale[3]; int32_t nt; if ((rc = get_rgb(S, ps, "scale", one, scale, &nt))) return rc;
            if (type == "point") {                                  /* make_point_light constructors.rs:330-337 */
                float I[3], from[3]; if ((rc = get_rgb(S, ps, "I", one, I, &nt))) return rc; vec3("from", zero, from);
                l.type = FTN_LIGHT_POINT; for (int k = 0; k < 3; k++) l.rgb[k] = I[k] * scale[k];
                ftn_transform_translate(from, &l.light_to_world); const float o[3] = {0, 0, 0}; ftn_transform_point(&l.light_to_world, o, l.v);
            } else if (type == "distant") {                         /* make_distant_light :320-328, DistantLight::new distant.rs:23-31 */
                float Lr[3], from[3], to[3]; if ((rc = get_rgb(S, ps, "L", one, Lr, &nt))) return rc; vec3("from", zero, from); vec3("to", zup, to);
                l.type = FTN_LIGHT_DISTANT; for (int k = 0; k < 3; k++) l.rgb[k] = Lr[k] * scale[k];
                const float d[3] = {from[0] - to[0], from[1] - to[1], from[2] - to[2]};
                const float mag = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]), inv = 1.0f / mag;
                for (int k = 0; k < 3; k++) l.v[k] = d[k] * inv;
                ftn_transform_identity(&l.light_to_world);
            } else if (type == "infinite") {                        /* make_infinite_area_light :339-359 */
                float Lr[3]; if ((rc = get_rgb(S, ps, "L", one, Lr, &nt))) return rc;
                l.type = FTN_LIGHT_INFINITE; l.light_to_world = S->tf.back();
                ftn_pbrt::Env env;
                auto mn = ps.find("mapname");
                if (mn != ps.end() && mn->second.type == "string" && !mn->second.s.empty()) {            /* load_mipmap (imageio/mod.rs:81-106): texel * scale[0], no gamma for .exr */
                    const std::string file = S->base_dir + "/" + mn->second.s[0];
                    if (file.size() < 4 || file.substr(file.size() - 4) != ".exr") return fail(S, FTN_ERR_UNSUPPORTED, "environment maps are read from OpenEXR files only");
                    if ((rc = ftn_exr_read(file.c_str(), &env.w, &env.h, nullptr))) return fail(S, rc, ftn_imageio_last_error());
                    env.texels.resize((size_t)env.w * env.h * 3);
                    if ((rc = ftn_exr_read(file.c_str(), &env.w, &env.h, env.texels.data()))) return fail(S, rc, ftn_imageio_last_error());
                    for (float& t : env.texels) t = t * scale[0];
                } else { env.w = env.h = 1; env.texels.assign(Lr, Lr + 3); }                             /* new_uniform: 1x1 map */
                S->env_store.push_back(env);
                l.envmap = (int)S->env_store.size() - 1;
            } else return fail(S, FTN_ERR_INVALID_ARGUMENT, "UnknownName(" + type + ")");
            S->lights.push_back(l);
        } else if (w == "Shape") {
            const GState& g = S->gs.back();
            if (type == "sphere") {                                 /* make_sphere constructors.rs:38-57 */
                const float radius = getf(ps, "radius", 1.0f);
                ftn_transform w2o; ftn_transform_inverse(&S->tf.back(), &w2o);
                ftn_sphere s; ftn_sphere_init(&S->tf.back(), &w2o, g.rev ? 1 : 0, radius, getf(ps, "zmin", -radius), getf(ps, "zmax", radius), getf(ps, "phimax", 360.0f), &s);
                S->spheres.push_back(s);
                ftn_prim p; p.shape_kind = FTN_SHAPE_SPHERE; p.shape_index = (uint32_t)S->spheres.size() - 1; p.material = g.material; p.area_emit = g.area; S->prims.push_back(p);
            } else if (type == "trianglemesh") {                    /* make_triangle_mesh :59-92 */
                Mesh m; auto it = ps.find("indices"); auto ip = ps.find("P");
                if (it == ps.end() || ip == ps.end()) return fail(S, FTN_ERR_INVALID_ARGUMENT, "trianglemesh needs indices and P");
                for (int v : it->second.i) m.idx.push_back((uint32_t)v);
                if (m.idx.size() % 3) return fail(S, FTN_ERR_INVALID_ARGUMENT, "indices is not a multiple of 3");
                m.P = ip->second.f;
                if (ps.find("N") != ps.end()) m.N = ps["N"].f;
                if (ps.find("S") != ps.end()) m.S = ps["S"].f;                 /* constructors.rs:63 */
                if (ps.find("uv") != ps.end()) m.UV = ps["uv"].f; else if (ps.find("st") != ps.end()) m.UV = ps["st"].f;
                if ((!m.N.empty() && m.N.size() != m.P.size()) || (!m.S.empty() && m.S.size() != m.P.size()) || (!m.UV.empty() && m.UV.size() / 2 != m.P.size() / 3)) return fail(S, FTN_ERR_INVALID_ARGUMENT, "per-vertex array length mismatch");
                if ((rc = add_mesh(S, m))) return rc;
            } else if (type == "plymesh") {
                auto it = ps.find("filename"); if (it == ps.end() || it->second.s.empty()) return fail(S, FTN_ERR_INVALID_ARGUMENT, "plymesh needs a filename");
                Mesh m; if ((rc = load_ply(S, S->base_dir + "/" + it->second.s[0], &m))) return rc;
                if ((rc = add_mesh(S, m))) return rc;
            } else return fail(S, FTN_ERR_INVALID_ARGUMENT, "UnknownName(" + type + ")");
        } else return fail(S, FTN_ERR_INVALID_ARGUMENT, "unknown directive " + w);
        (void)d05;
    }
    if (!world) return fail(S, FTN_ERR_INVALID_ARGUMENT, "no WorldBegin");
    /* header objects: make_camera / make_sampler / make_film (pbrt.rs:426-505) */
    const int xres = geti(S->film_params, "xresolution", 640), yres = geti(S->film_params, "yresolution", 480);
    { auto it = S->film_params.find("filename"); if (it != S->film_params.end() && !it->second.s.empty()) S->film_name = it->second.s[0]; }
    float crop[4] = {0.0f, 0.0f, 1.0f, 1.0f};
    { auto it = S->film_params.find("cropwindow"); if (it != S->film_params.end() && it->second.f.size() >= 4) { crop[0] = it->second.f[0]; crop[2] = it->second.f[1]; crop[1] = it->second.f[2]; crop[3] = it->second.f[3]; } }
    const int32_t res[2] = {xres, yres};
    ftn_film_init(res, crop, &S->film);
    S->spp = geti(S->sampler_params, "pixelsamples", 16);
    if (!S->has_camera) return fail(S, FTN_ERR_INVALID_ARGUMENT, "no Camera statement");
    if (S->camera_params["name"].s.empty() || S->camera_params["name"].s[0] != "perspective") return fail(S, FTN_ERR_INVALID_ARGUMENT, "UnknownName(camera)");
    ftn_transform c2w; ftn_transform_inverse(&S->camera_tf, &c2w);
    const float aspect = getf(S->camera_params, "frameaspectratio", (float)xres / (float)yres);
    float sw[4];
    if (aspect > 1.0f) { sw[0] = -aspect; sw[1] = -1.0f; sw[2] = aspect; sw[3] = 1.0f; } else { sw[0] = -1.0f; sw[1] = -1.0f / aspect; sw[2] = 1.0f; sw[3] = 1.0f / aspect; }
    const float sh[2] = {getf(S->camera_params, "shutteropen", 0.0f), getf(S->camera_params, "shutterclose", 1.0f)};
    if ((rc = ftn_camera_perspective(&c2w, res, sw, sh, getf(S->camera_params, "lensradius", 0.0f), getf(S->camera_params, "focaldistance", 1e6f), getf(S->camera_params, "fov", 90.0f), &S->camera)))
        return fail(S, rc, ftn_last_error());
    /* flat description */
    ftn_scene_desc& d = S->desc; memset(&d, 0, sizeof(d));
    for (auto& e : S->env_store) { ftn_envmap m; m.width = e.w; m.height = e.h; m.texels = e.texels.data(); S->envmaps.push_back(m); }
    d.n_prims = (uint32_t)S->prims.size(); d.prims = S->prims.data();
    d.n_triangles = (uint32_t)(S->tri_indices.size() / 3); d.tri_indices = S->tri_indices.data(); d.tri_mesh = S->tri_mesh.data();
    d.n_vertices = (uint32_t)(S->P.size() / 3); d.P = S->P.data(); d.N = S->any_n ? S->N.data() : nullptr; d.UV = S->any_uv ? S->UV.data() : nullptr; d.S = S->any_t ? S->T.data() : nullptr;
    d.n_meshes = (uint32_t)S->meshes.size(); d.meshes = S->meshes.data();
    d.n_spheres = (uint32_t)S->spheres.size(); d.spheres = S->spheres.data();
    d.n_materials = (uint32_t)S->materials.size(); d.materials = S->materials.data();
    d.n_area_emit = (uint32_t)(S->area_emit.size() / 3); d.area_emit = S->area_emit.data();
    d.n_lights = (uint32_t)S->lights.size(); d.lights = S->lights.data();
    d.n_envmaps = (uint32_t)S->envmaps.size(); d.envmaps = S->envmaps.data();
    if (!S->textures.empty()) {
        for (auto& im : S->image_store) { ftn_image i; i.width = im.w; i.height = im.h; i.wrap = im.wrap; i._pad = 0; i.texels = im.texels.data(); S->images.push_back(i); }
        d.n_textures = (uint32_t)S->textures.size(); d.textures = S->textures.data(); d.material_textures = S->mtex.data();
        d.n_images = (uint32_t)S->images.size(); d.images = S->images.data();
    }
    return FTN_OK;
}

thread_local std::string g_pbrt_err;

}  // namespace

extern "C" {

int ftn_pbrt_load(const char* path, ftn_pbrt** out) {
    if (!path || !out) return FTN_ERR_INVALID_ARGUMENT;
    ftn_pbrt* S = new ftn_pbrt();
    std::string p(path); size_t sl = p.find_last_of('/');
    S->base_dir = sl == std::string::npos ? "." : p.substr(0, sl);
    int rc = parse(S, p);
    if (rc) { g_pbrt_err = S->error; delete S; return rc; }
    *out = S; return FTN_OK;
}
void ftn_pbrt_destroy(ftn_pbrt* s) { delete s; }
const ftn_scene_desc* ftn_pbrt_scene(const ftn_pbrt* s) { return &s->desc; }
const ftn_camera_desc* ftn_pbrt_camera(const ftn_pbrt* s) { return &s->camera; }
const ftn_film_desc* ftn_pbrt_film(const ftn_pbrt* s) { return &s->film; }
int ftn_pbrt_samples_per_pixel(const ftn_pbrt* s) { return s->spp; }
const char* ftn_pbrt_film_name(const ftn_pbrt* s) { return s->film_name.c_str(); }
const char* ftn_pbrt_last_error(void) { return g_pbrt_err.c_str(); }
/* PLY reader on its own (constructors.rs:94-190): fills caller arrays after a sizing call with NULL outputs */
int ftn_ply_load(const char* path, uint32_t* n_vertices, uint32_t* n_triangles, float* P, float* N, float* UV, uint32_t* indices, int* has_normals, int* has_uvs) {
    ftn_pbrt tmp; Mesh m;
    int rc = load_ply(&tmp, path, &m);
    if (rc) { g_pbrt_err = tmp.error; return rc; }
    if (n_vertices) *n_vertices = (uint32_t)(m.P.size() / 3);
    if (n_triangles) *n_triangles = (uint32_t)(m.idx.size() / 3);
    if (has_normals) *has_normals = m.N.empty() ? 0 : 1;
    if (has_uvs) *has_uvs = m.UV.empty() ? 0 : 1;
    if (P) memcpy(P, m.P.data(), m.P.size() * 4);
    if (N && !m.N.empty()) memcpy(N, m.N.data(), m.N.size() * 4);
    if (UV && !m.UV.empty()) memcpy(UV, m.UV.data(), m.UV.size() * 4);
    if (indices) memcpy(indices, m.idx.data(), m.idx.size() * 4);
    return FTN_OK;
}

}  // extern "C"
