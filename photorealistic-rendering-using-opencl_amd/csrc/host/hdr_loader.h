// hdr_loader.h -- Radiance RGBE (.hdr / .pic) reader for the `-hdr <file>` environment map.
// Replaces loadHDR (include/Texture/texture.h:31-39 -> stbi_loadf) of the reference: same pixel values
// (mantissa * 2^(exponent - 136), an all-zero exponent byte = black), rows in file order, 3 floats per pixel --
// what the reference hands to glTexImage2D(GL_RGB32F) (include/GL/cl_gl_interop.h:71-86).
#pragma once
#include <string>
#include <vector>

namespace prt {
namespace IO {

// true on success.  Supports what the reference's loader supports: "#?RADIANCE" / "#?RGBE" files with
// FORMAT=32-bit_rle_rgbe and the standard "-Y <h> +X <w>" orientation, flat or run-length-encoded scanlines.
bool load_hdr(const std::string& path, std::vector<float>& rgb, int& width, int& height, std::string& err);
// the same from memory (tests)
bool decode_hdr(const unsigned char* data, size_t size, std::vector<float>& rgb, int& width, int& height, std::string& err);

// `-encoder 1` of the reference (saveImage -> stbi_write_hdr("render.hdr", w, h, 3, pixels), include/GL/cl_gl_interop.h:151-156):
// `channels` floats per pixel (the first three are written), rows of `pixels` bottom-up (the framebuffer's order: the reference flips
// on write, :139) or top-down; the file is top-down, run-length encoded.
bool write_hdr(const std::string& path, const float* pixels, int width, int height, int channels, bool bottom_up, std::string& err);

}  // namespace IO
}  // namespace prt
