// hdr_loader.cpp -- see hdr_loader.h.  Written from the Radiance picture format (Ward, "Real Pixels", Graphics
// Gems II; the RGBE header / scanline conventions), not from the reference's vendored decoder.
#include "hdr_loader.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace prt {
namespace IO {

namespace {

struct Reader {
    const unsigned char* p;
    size_t n, pos;
    bool eof() const { return pos >= n; }
    int get() { return pos < n ? p[pos++] : -1; }
    bool line(std::string& out) {                 // header lines end with '\n'
        out.clear();
        if (eof()) return false;
        while (!eof()) {
            const char ch = (char)p[pos++];
            if (ch == '\n') break;
            if (out.size() < 1024) out.push_back(ch);
        }
        return true;
    }
};

inline void rgbe_to_float(const unsigned char* q, float* out) {
    if (q[3] == 0) { out[0] = out[1] = out[2] = 0.0f; return; }
    const float f = std::ldexp(1.0f, (int)q[3] - (128 + 8));
    out[0] = (float)q[0] * f; out[1] = (float)q[1] * f; out[2] = (float)q[2] * f;
}

}  // namespace

bool decode_hdr(const unsigned char* data, size_t size, std::vector<float>& rgb, int& width, int& height, std::string& err) {
    Reader r{data, size, 0};
    std::string ln;
    if (!r.line(ln) || (ln != "#?RADIANCE" && ln != "#?RGBE")) { err = "not a Radiance picture (no #?RADIANCE / #?RGBE signature)"; return false; }
    bool format_ok = false;
    for (;;) {
        if (!r.line(ln)) { err = "truncated header"; return false; }
        if (ln.empty()) break;
        if (ln == "FORMAT=32-bit_rle_rgbe") format_ok = true;
    }
    if (!format_ok) { err = "unsupported pixel format (FORMAT=32-bit_rle_rgbe expected)"; return false; }
    if (!r.line(ln)) { err = "missing resolution line"; return false; }
    long h = 0, w = 0;
    {
        const char* s = ln.c_str();
        char* end = nullptr;
        if (std::strncmp(s, "-Y ", 3) != 0) { err = "unsupported orientation (only -Y <height> +X <width>)"; return false; }
        h = std::strtol(s + 3, &end, 10);
        while (*end == ' ') ++end;
        if (std::strncmp(end, "+X ", 3) != 0) { err = "unsupported orientation (only -Y <height> +X <width>)"; return false; }
        w = std::strtol(end + 3, nullptr, 10);
    }
    if (w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24) || (unsigned long long)w * (unsigned long long)h > (1ull << 28)) {
        err = "unreasonable picture size";
        return false;
    }
    // The decoded picture is allocated only if the bytes that are left can encode it: flat data needs 4 bytes per pixel, a
    // run-length scanline its 4-byte header and, per channel, 2 bytes for every run of at most 127 pixels -- so the memory a
    // file can claim is bounded by a fixed multiple (~190 x) of its own size.
    {
        const bool rle_possible = !(w < 8 || w >= 32768);
        const unsigned long long flat_bytes = 4ull * (unsigned long long)w * (unsigned long long)h;
        const unsigned long long rle_bytes = (unsigned long long)h * (4ull + 8ull * (unsigned long long)((w + 126) / 127));
        const unsigned long long need = (rle_possible && rle_bytes < flat_bytes) ? rle_bytes : flat_bytes;
        if ((unsigned long long)(size - r.pos) < need) { err = "truncated pixel data"; return false; }
    }
    width = (int)w; height = (int)h;
    rgb.assign((size_t)w * h * 3, 0.0f);
    std::vector<unsigned char> scan((size_t)w * 4);
    auto flat_from = [&](size_t first_pixel, const unsigned char* head, int head_bytes) -> bool {
        // uncompressed RGBE quadruples from pixel `first_pixel` on; `head` = bytes of it already consumed
        size_t px = first_pixel;
        unsigned char q[4];
        int have = head_bytes;
        for (int k = 0; k < head_bytes; ++k) q[k] = head[k];
        const size_t total = (size_t)w * h;
        while (px < total) {
            while (have < 4) { const int c = r.get(); if (c < 0) { err = "truncated pixel data"; return false; } q[have++] = (unsigned char)c; }
            rgbe_to_float(q, &rgb[px * 3]);
            ++px; have = 0;
        }
        return true;
    };
    if (w < 8 || w >= 32768) return flat_from(0, nullptr, 0);
    for (long j = 0; j < h; ++j) {
        unsigned char hd[4];
        for (int k = 0; k < 4; ++k) { const int c = r.get(); if (c < 0) { err = "truncated pixel data"; return false; } hd[k] = (unsigned char)c; }
        if (hd[0] != 2 || hd[1] != 2 || (hd[2] & 0x80)) {
            // not run-length encoded: the four bytes are the first pixel of flat data.  Only the first scanline can say so; a
            // later one without the run-length header is a corrupt file (the reference's stb loader would restart the whole
            // picture as flat data from there; refused here)
            if (j != 0) { err = "corrupt run-length data (scanline without a run-length header)"; return false; }
            if ((unsigned long long)(size - r.pos) + 4ull < 4ull * (unsigned long long)w * (unsigned long long)h) { err = "truncated pixel data"; return false; }
            return flat_from(0, hd, 4);
        }
        if ((((long)hd[2]) << 8 | hd[3]) != w) { err = "corrupt run-length scanline (width mismatch)"; return false; }
        for (int ch = 0; ch < 4; ++ch) {
            long i = 0;
            while (i < w) {
                int count = r.get();
                if (count < 0) { err = "truncated pixel data"; return false; }
                if (count > 128) {                  // a run
                    count -= 128;
                    const int v = r.get();
                    if (v < 0) { err = "truncated pixel data"; return false; }
                    if (count == 0 || i + count > w) { err = "corrupt run-length scanline"; return false; }
                    for (int k = 0; k < count; ++k) scan[(size_t)(i++) * 4 + ch] = (unsigned char)v;
                } else {                            // literals
                    if (count == 0 || i + count > w) { err = "corrupt run-length scanline"; return false; }
                    for (int k = 0; k < count; ++k) {
                        const int v = r.get();
                        if (v < 0) { err = "truncated pixel data"; return false; }
                        scan[(size_t)(i++) * 4 + ch] = (unsigned char)v;
                    }
                }
            }
        }
        for (long i = 0; i < w; ++i) rgbe_to_float(&scan[(size_t)i * 4], &rgb[((size_t)j * w + i) * 3]);
    }
    return true;
}

bool load_hdr(const std::string& path, std::vector<float>& rgb, int& width, int& height, std::string& err) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) {
        buf.insert(buf.end(), tmp, tmp + n);
        if (buf.size() > (1ull << 31)) { std::fclose(f); err = path + ": file too large"; return false; }
    }
    std::fclose(f);
    if (!decode_hdr(buf.data(), buf.size(), rgb, width, height, err)) { err = path + ": " + err; return false; }
    return true;
}

// ---- writer: `-encoder 1` of the reference (saveImage, include/GL/cl_gl_interop.h:151-156 -> stbi_write_hdr) -------------------
// Radiance RGBE as the format defines it: the largest component m of a pixel fixes the shared exponent e with 2^(e-1) <= m < 2^e,
// each component is stored as floor(c * 256 / 2^e) and the exponent byte as e + 128; a pixel whose largest component is below
// 1e-32 is four zero bytes.  Scanlines are run-length encoded per channel (widths 8 .. 32767), rows top to bottom.
namespace {
void float_to_rgbe(const float* c, unsigned char* q) {
    const float m = std::fmax(c[0], std::fmax(c[1], c[2]));
    if (!(m >= 1e-32f)) { q[0] = q[1] = q[2] = q[3] = 0; return; }                 // (a NaN or negative picture value ends here too)
    int e = 0;
    const float scale = std::frexp(m, &e) * 256.0f / m;
    for (int k = 0; k < 3; ++k) { const float v = c[k] * scale; q[k] = (unsigned char)(v > 0.0f ? (v < 255.0f ? (int)v : 255) : 0); }
    q[3] = (unsigned char)(e + 128);
}
void rle_channel(const unsigned char* row, int n, int stride, std::vector<unsigned char>& out) {
    int i = 0;
    while (i < n) {
        int j = i;
        while (j < n && j - i < 127 && row[(size_t)j * stride] == row[(size_t)i * stride]) ++j;
        if (j - i >= 3) { out.push_back((unsigned char)(128 + (j - i))); out.push_back(row[(size_t)i * stride]); i = j; continue; }
        int k = i;                                                  // literals up to the next run of three
        while (k < n && k - i < 128 &&
               !(k + 2 < n && row[(size_t)k * stride] == row[(size_t)(k + 1) * stride] && row[(size_t)k * stride] == row[(size_t)(k + 2) * stride])) ++k;
        out.push_back((unsigned char)(k - i));
        for (int t = i; t < k; ++t) out.push_back(row[(size_t)t * stride]);
        i = k;
    }
}
}  // namespace

bool write_hdr(const std::string& path, const float* pixels, int width, int height, int channels, bool bottom_up, std::string& err) {
    if (!pixels || width <= 0 || height <= 0 || channels < 3) { err = "write_hdr: bad arguments"; return false; }
    std::vector<unsigned char> out;
    const std::string head = "#?RADIANCE\n# written by libprt (prt_render -encoder 1)\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y " +
                             std::to_string(height) + " +X " + std::to_string(width) + "\n";
    out.insert(out.end(), head.begin(), head.end());
    std::vector<unsigned char> scan((size_t)width * 4);
    for (int j = 0; j < height; ++j) {
        const float* row = pixels + (size_t)(bottom_up ? height - 1 - j : j) * width * channels;
        for (int i = 0; i < width; ++i) float_to_rgbe(row + (size_t)i * channels, &scan[(size_t)i * 4]);
        if (width < 8 || width >= 32768) { out.insert(out.end(), scan.begin(), scan.end()); continue; }
        out.push_back(2); out.push_back(2); out.push_back((unsigned char)(width >> 8)); out.push_back((unsigned char)(width & 255));
        for (int ch = 0; ch < 4; ++ch) rle_channel(scan.data() + ch, width, 4, out);
    }
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { err = "cannot open " + path + " for writing"; return false; }
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    if (std::fclose(f) != 0 || !ok) { err = "cannot write " + path; return false; }
    return true;
}

}  // namespace IO
}  // namespace prt
