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
    if ((size - r.pos) < (size_t)h * 4) { err = "truncated pixel data"; return false; }      // cheapest possible encoding still needs this
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
            // not run-length encoded: the four bytes are the first pixel of flat data (only legal on the first scanline;
            // later on it would be a corrupt file, which decodes the same way the reference's loader does)
            return flat_from((size_t)j * w, hd, 4);
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

}  // namespace IO
}  // namespace prt
