/*
 * ftn_imageio.cpp -- the output stage's file format: OpenEXR scanline images with 32-bit float R, G, B channels, as
 * write_exr / read_exr produce and consume (src/imageio/exr.rs:11-87).  The reference delegates the container to the `exr`
 * crate (RLE-compressed scanline blocks); this writer emits the same layer (name "image", channels B G R, FLOAT, increasing
 * line order) with NO_COMPRESSION, which every OpenEXR reader accepts and which keeps the sample bits exactly as rendered.
 * The reader handles NO_COMPRESSION, RLE, ZIPS and ZIP scanline files with FLOAT or HALF channels (what the writer above and the
 * reference's writer produce, plus the usual encoding of environment maps and textures found in PBRT scenes; zlib inflates).
 */
#include "../../include/fountain_hip.h"
#include "detmath.h"

#include <algorithm>
#include <cstdio>
#include <zlib.h>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_io_err;
int io_fail(int code, const std::string& m) { g_io_err = m; return code; }

void put_i32(std::vector<unsigned char>& b, int32_t v) { unsigned char t[4]; memcpy(t, &v, 4); b.insert(b.end(), t, t + 4); }
void put_f32(std::vector<unsigned char>& b, float v) { unsigned char t[4]; memcpy(t, &v, 4); b.insert(b.end(), t, t + 4); }
void put_str(std::vector<unsigned char>& b, const char* s) { b.insert(b.end(), s, s + strlen(s) + 1); }
void put_attr(std::vector<unsigned char>& b, const char* name, const char* type, const std::vector<unsigned char>& data) {
    put_str(b, name); put_str(b, type); put_i32(b, (int32_t)data.size()); b.insert(b.end(), data.begin(), data.end());
}

float half_to_float(uint16_t h) {
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = s;
        else { int sh = 0; uint32_t mm = m; while (!(mm & 1024u)) { mm <<= 1; sh++; } u = s | ((uint32_t)(113 - sh) << 23) | ((mm & 1023u) << 13); }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

/* the byte predictor and the two-halves interleave shared by OpenEXR's RLE and ZIP blocks */
void unpredict(std::vector<unsigned char>& tmp, std::vector<unsigned char>& out) {
    const size_t n = tmp.size();
    for (size_t i = 1; i < n; i++) tmp[i] = (unsigned char)(tmp[i - 1] + tmp[i] - 128);
    out.resize(n);
    const size_t half = (n + 1) / 2;
    for (size_t i = 0, a = 0, b = half; i < n;) { out[i++] = tmp[a++]; if (i < n) out[i++] = tmp[b++]; }
}
/* OpenEXR RLE block: run-length bytes, then predictor + interleave undone */
bool rle_decode(const unsigned char* in, size_t n_in, std::vector<unsigned char>& out, size_t n_out) {
    std::vector<unsigned char> tmp; tmp.reserve(n_out);
    size_t p = 0;
    while (p < n_in) {
        const int c = (signed char)in[p++];
        if (c < 0) { const size_t k = (size_t)(-c); if (p + k > n_in) return false; tmp.insert(tmp.end(), in + p, in + p + k); p += k; }
        else { if (p >= n_in) return false; tmp.insert(tmp.end(), (size_t)c + 1, in[p]); p++; }
    }
    if (tmp.size() != n_out) return false;
    unpredict(tmp, out);
    return true;
}
bool zip_decode(const unsigned char* in, size_t n_in, std::vector<unsigned char>& out, size_t n_out) {
    std::vector<unsigned char> tmp(n_out);
    uLongf got = (uLongf)n_out;
    if (uncompress(tmp.data(), &got, in, (uLong)n_in) != Z_OK || got != n_out) return false;
    unpredict(tmp, out);
    return true;
}

}  // namespace

extern "C" {

const char* ftn_imageio_last_error(void) { return g_io_err.c_str(); }

/* write_exr (src/imageio/exr.rs:47-87): rgb = 3 floats per pixel, row-major, w*h pixels */
int ftn_exr_write(const char* path, const float* rgb, uint32_t w, uint32_t h) {
    if (!path || !rgb || w == 0 || h == 0) return io_fail(FTN_ERR_INVALID_ARGUMENT, "ftn_exr_write: bad arguments");
    std::vector<unsigned char> hd;
    const unsigned char magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};
    hd.insert(hd.end(), magic, magic + 8);
    std::vector<unsigned char> d;
    for (const char* ch : {"B", "G", "R"}) { put_str(d, ch); put_i32(d, 2 /* FLOAT */); d.push_back(1 /* pLinear */); d.push_back(0); d.push_back(0); d.push_back(0); put_i32(d, 1); put_i32(d, 1); }
    d.push_back(0);
    put_attr(hd, "channels", "chlist", d);
    put_attr(hd, "compression", "compression", {0});
    d.clear(); put_i32(d, 0); put_i32(d, 0); put_i32(d, (int32_t)w - 1); put_i32(d, (int32_t)h - 1);
    put_attr(hd, "dataWindow", "box2i", d);
    put_attr(hd, "displayWindow", "box2i", d);
    put_attr(hd, "lineOrder", "lineOrder", {0});
    d.clear(); put_str(d, "image"); d.pop_back();
    put_attr(hd, "name", "string", d);
    d.clear(); put_f32(d, 1.0f); put_attr(hd, "pixelAspectRatio", "float", d);
    d.clear(); put_f32(d, 0.0f); put_f32(d, 0.0f); put_attr(hd, "screenWindowCenter", "v2f", d);
    d.clear(); put_f32(d, 1.0f); put_attr(hd, "screenWindowWidth", "float", d);
    hd.push_back(0);
    FILE* f = fopen(path, "wb");
    if (!f) return io_fail(FTN_ERR_INVALID_ARGUMENT, std::string("cannot create ") + path);
    const size_t row_bytes = (size_t)w * 12, block = 8 + row_bytes;
    std::vector<uint64_t> offs(h);
    for (uint32_t y = 0; y < h; y++) offs[y] = hd.size() + (size_t)h * 8 + (size_t)y * block;
    bool ok = fwrite(hd.data(), 1, hd.size(), f) == hd.size() && fwrite(offs.data(), 8, h, f) == h;
    std::vector<float> row((size_t)w * 3);
    for (uint32_t y = 0; y < h && ok; y++) {
        const float* src = rgb + (size_t)y * w * 3;
        for (uint32_t x = 0; x < w; x++) { row[x] = src[3 * x + 2]; row[w + x] = src[3 * x + 1]; row[2 * (size_t)w + x] = src[3 * x]; }
        const int32_t hdr2[2] = {(int32_t)y, (int32_t)row_bytes};
        ok = fwrite(hdr2, 4, 2, f) == 2 && fwrite(row.data(), 4, row.size(), f) == row.size();
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? FTN_OK : io_fail(FTN_ERR_INTERNAL, std::string("short write to ") + path);
}

/* read_exr (src/imageio/exr.rs:11-45): first call with rgb_out == NULL for the size */
int ftn_exr_read(const char* path, uint32_t* w_out, uint32_t* h_out, float* rgb_out) {
    if (!path) return io_fail(FTN_ERR_INVALID_ARGUMENT, "ftn_exr_read: bad arguments");
    FILE* f = fopen(path, "rb");
    if (!f) return io_fail(FTN_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
    std::vector<unsigned char> buf;
    { unsigned char tmp[65536]; size_t n; while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n); }
    fclose(f);
    if (buf.size() < 8 || buf[0] != 0x76 || buf[1] != 0x2f || buf[2] != 0x31 || buf[3] != 0x01) return io_fail(FTN_ERR_INVALID_ARGUMENT, std::string(path) + ": not an OpenEXR file");
    if (buf[5] & 0x1a) return io_fail(FTN_ERR_UNSUPPORTED, "tiled / deep / multi-part OpenEXR files are not supported");
    size_t p = 8;
    struct Chan { std::string name; int type; };
    std::vector<Chan> chans; int compression = -1, line_order = 0; int32_t dw[4] = {0, 0, -1, -1};
    auto cstr = [&](std::string* s) { size_t e = p; while (e < buf.size() && buf[e]) e++; if (e >= buf.size()) return false; s->assign((const char*)&buf[p], e - p); p = e + 1; return true; };
    for (;;) {
        std::string name, type;
        if (!cstr(&name)) return io_fail(FTN_ERR_INVALID_ARGUMENT, "truncated OpenEXR header");
        if (name.empty()) break;
        if (!cstr(&type) || p + 4 > buf.size()) return io_fail(FTN_ERR_INVALID_ARGUMENT, "truncated OpenEXR header");
        int32_t size; memcpy(&size, &buf[p], 4); p += 4;
        if (size < 0 || p + (size_t)size > buf.size()) return io_fail(FTN_ERR_INVALID_ARGUMENT, "truncated OpenEXR header");
        const size_t a = p; p += (size_t)size;
        if (name == "channels") {
            size_t q = a;
            while (q < a + size && buf[q]) { Chan c; while (buf[q]) c.name.push_back((char)buf[q++]); q++; int32_t t; memcpy(&t, &buf[q], 4); c.type = t; q += 16; chans.push_back(c); }
        } else if (name == "compression") compression = buf[a];
        else if (name == "dataWindow") memcpy(dw, &buf[a], 16);
        else if (name == "lineOrder") line_order = buf[a];
    }
    const int64_t w = (int64_t)dw[2] - dw[0] + 1, h = (int64_t)dw[3] - dw[1] + 1;
    if (w <= 0 || h <= 0) return io_fail(FTN_ERR_INVALID_ARGUMENT, "OpenEXR file has no dataWindow");
    if (compression < 0 || compression > 3) return io_fail(FTN_ERR_UNSUPPORTED, "only NO_COMPRESSION, RLE, ZIPS and ZIP OpenEXR scanline files are supported (not PIZ / PXR24 / B44 / DWA)");
    const int64_t lines_per_block = compression == 3 ? 16 : 1;
    if (w_out) *w_out = (uint32_t)w;
    if (h_out) *h_out = (uint32_t)h;
    if (!rgb_out) return FTN_OK;
    int ci[3] = {-1, -1, -1}; size_t row_bytes = 0; std::vector<size_t> ch_off(chans.size());
    for (size_t c = 0; c < chans.size(); c++) {
        ch_off[c] = row_bytes;
        if (chans[c].type != 1 && chans[c].type != 2) return io_fail(FTN_ERR_UNSUPPORTED, "UINT OpenEXR channels are not supported");   /* read_exr panics */
        row_bytes += (size_t)w * (chans[c].type == 2 ? 4 : 2);
        if (chans[c].name == "R") ci[0] = (int)c; else if (chans[c].name == "G") ci[1] = (int)c; else if (chans[c].name == "B") ci[2] = (int)c;
    }
    if (ci[0] < 0 || ci[1] < 0 || ci[2] < 0) return io_fail(FTN_ERR_INVALID_ARGUMENT, "OpenEXR file lacks R, G or B");
    (void)line_order;                                       /* every block carries its own y */
    std::vector<unsigned char> raw;
    const int64_t n_blocks = (h + lines_per_block - 1) / lines_per_block;
    if (p + (size_t)n_blocks * 8 > buf.size()) return io_fail(FTN_ERR_INVALID_ARGUMENT, "truncated OpenEXR offset table");
    for (int64_t b = 0; b < n_blocks; b++) {
        uint64_t off; memcpy(&off, &buf[p + (size_t)b * 8], 8);
        if (off + 8 > buf.size()) return io_fail(FTN_ERR_INVALID_ARGUMENT, "truncated OpenEXR scanline");
        int32_t y, n; memcpy(&y, &buf[off], 4); memcpy(&n, &buf[off + 4], 4);
        if (n < 0 || off + 8 + (size_t)n > buf.size() || y < dw[1] || y > dw[3]) return io_fail(FTN_ERR_INVALID_ARGUMENT, "corrupt OpenEXR scanline");
        const int64_t lines = std::min<int64_t>(lines_per_block, (int64_t)dw[3] - y + 1);
        const size_t raw_bytes = row_bytes * (size_t)lines;
        const unsigned char* data = &buf[off + 8];
        if (compression != 0 && (size_t)n < raw_bytes) {          /* a block that does not shrink is stored raw */
            const bool ok = compression == 1 ? rle_decode(data, (size_t)n, raw, raw_bytes) : zip_decode(data, (size_t)n, raw, raw_bytes);
            if (!ok) return io_fail(FTN_ERR_INVALID_ARGUMENT, "corrupt compressed OpenEXR block");
            data = raw.data();
        } else if ((size_t)n != raw_bytes) return io_fail(FTN_ERR_INVALID_ARGUMENT, "unexpected OpenEXR block size");
        for (int64_t ly = 0; ly < lines; ly++) {
            float* dst = rgb_out + (size_t)(y + ly - dw[1]) * (size_t)w * 3;
            for (int k = 0; k < 3; k++) {
                const unsigned char* src = data + (size_t)ly * row_bytes + ch_off[ci[k]];
                if (chans[ci[k]].type == 2) for (int64_t x = 0; x < w; x++) memcpy(&dst[3 * x + k], src + 4 * x, 4);
                else for (int64_t x = 0; x < w; x++) { uint16_t hv; memcpy(&hv, src + 2 * x, 2); dst[3 * x + k] = half_to_float(hv); }
            }
        }
    }
    return FTN_OK;
}

}  // extern "C"

/* load_mipmap's gamma step: imageio/mod.rs:101-107 with inverse_gamma_correct (:169-175); f32::powf through the deterministic powf
 * (detmath.h), like every other transcendental of this library */
int ftn_image_inverse_gamma(float* texels, size_t n) {
    if (!texels && n) return io_fail(FTN_ERR_INVALID_ARGUMENT, "null texel array");
    for (size_t i = 0; i < n; i++) {
        const float v = texels[i];
        texels[i] = v <= 0.04045f ? v * 1.0f / 12.92f : ftn_det::powf_det((v + 0.055f) * 1.0f / 1.055f, 2.4f);
    }
    return FTN_OK;
}
