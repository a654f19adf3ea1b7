// exr_writer.hpp -- minimal OpenEXR 2 scanline writer (RGB, float32 or float16; ZIP-compressed like the files the reference writes, or
// uncompressed) so that MultiFilm's DEFAULT output (`fileFormat=openexr`, `componentFormat=float16`;
// /root/reference/src/films/multifilm.cpp:104-117,200-205) needs no OpenEXR library.  File layout per the OpenEXR file-layout specification:
// magic, version, attribute list, chunk offset table, chunks with the channels of each scanline stored planar in alphabetical order
// (B, G, R).  Bitmap::writeOpenEXR builds `Imf::Header header(w, h)` (src/libcore/bitmap.cpp:3197), whose defaults are ZIP_COMPRESSION
// (blocks of 16 scanlines: byte reordering into even / odd halves, a delta predictor, then zlib) and INCREASING_Y: the default here too.
// Also the Radiance RGBE writer of `fileFormat=rgbe` (multifilm.cpp:119-125): flat (non run-length) scanlines, a valid .rgbe / .hdr file.
#pragma once
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include <cmath>
#include <zlib.h>

namespace gdpt {

inline uint16_t float_to_half(float f)
{ // IEEE binary32 -> binary16, round to nearest even, overflow to inf, subnormals kept
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (x > 0x7F800000u ? 0x200u : 0));      // inf / nan
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                          // rounds past the largest half
    if (x < 0x38800000u) {                                                                           // subnormal half or zero
        if (x < 0x33000000u) return (uint16_t)sign;
        const int e = (int)(x >> 23);
        uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
        const int shift = 126 - e;                                                                   // 14..24
        const uint32_t rnd = (m >> (shift - 1)) & 1u, sticky = (m & ((1u << (shift - 1)) - 1)) != 0;
        uint32_t h = m >> shift;
        if (rnd && (sticky || (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((x - 0x38000000u) >> 13);
    const uint32_t rem = x & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

class ExrWriter {
public:
    /// rgb: row-major [height][width][3] float32, top row first.  half = componentFormat float16.
    enum { NO_COMPRESSION = 0, ZIPS_COMPRESSION = 2, ZIP_COMPRESSION = 3 };
    static bool write(const std::string &path, const float *rgb, int width, int height, bool half, const std::string &log = "", int compression = ZIP_COMPRESSION)
    {
        if (compression != NO_COMPRESSION) return writeZip(path, rgb, width, height, half, log, compression);
        std::vector<char> hdr;
        put32(hdr, 20000630);       // magic
        put32(hdr, 2);              // version 2, single-part scanline
        {   // channels: B, G, R
            std::vector<char> ch;
            for (const char *nm : {"B", "G", "R"}) {
                ch.push_back(nm[0]); ch.push_back(0);
                put32(ch, half ? 1 : 2);                            // pixel type: 1 HALF, 2 FLOAT
                ch.push_back(0); ch.push_back(0); ch.push_back(0); ch.push_back(0);   // pLinear + reserved
                put32(ch, 1); put32(ch, 1);                         // x/y sampling
            }
            ch.push_back(0);
            attr(hdr, "channels", "chlist", ch);
        }
        attr(hdr, "compression", "compression", std::vector<char>(1, 0));       // NO_COMPRESSION
        { std::vector<char> b; put32(b, 0); put32(b, 0); put32(b, width - 1); put32(b, height - 1); attr(hdr, "dataWindow", "box2i", b); attr(hdr, "displayWindow", "box2i", b); }
        { const std::string g = "gdpt-mi355x (gradient-domain path tracer, HIP)"; attr(hdr, "generatedBy", "string", std::vector<char>(g.begin(), g.end())); }
        attr(hdr, "lineOrder", "lineOrder", std::vector<char>(1, 0));           // INCREASING_Y
        if (!log.empty()) attr(hdr, "log", "string", std::vector<char>(log.begin(), log.end()));     // multifilm.cpp:478-506 attaches the log
        { std::vector<char> f; putf(f, 1.0f); attr(hdr, "pixelAspectRatio", "float", f); }
        { std::vector<char> v; putf(v, 0.0f); putf(v, 0.0f); attr(hdr, "screenWindowCenter", "v2f", v); }
        { std::vector<char> f; putf(f, 1.0f); attr(hdr, "screenWindowWidth", "float", f); }
        hdr.push_back(0);           // end of header

        const size_t bpc = half ? 2 : 4, lineBytes = (size_t)width * 3 * bpc, chunk = 8 + lineBytes;
        std::ofstream f(path, std::ios::binary);
        if (!f) return false;
        f.write(hdr.data(), hdr.size());
        uint64_t off = hdr.size() + (uint64_t)height * 8;
        for (int y = 0; y < height; ++y, off += chunk) f.write(reinterpret_cast<const char *>(&off), 8);
        std::vector<char> line(lineBytes);
        for (int y = 0; y < height; ++y) {
            const int32_t yy = y, sz = (int32_t)lineBytes;
            f.write(reinterpret_cast<const char *>(&yy), 4);
            f.write(reinterpret_cast<const char *>(&sz), 4);
            for (int c = 0; c < 3; ++c) {                           // file order B, G, R  <-  memory order R, G, B
                const int src = 2 - c;
                for (int x = 0; x < width; ++x) {
                    const float v = rgb[((size_t)y * width + x) * 3 + src];
                    if (half) { const uint16_t h = float_to_half(v); std::memcpy(&line[((size_t)c * width + x) * 2], &h, 2); }
                    else std::memcpy(&line[((size_t)c * width + x) * 4], &v, 4);
                }
            }
            f.write(line.data(), lineBytes);
        }
        return (bool)f;
    }

    /// ZIP_COMPRESSION (16 scanlines per chunk) / ZIPS_COMPRESSION (1): ImfZip.cpp's reorder + predictor + zlib; a chunk that does not shrink is stored raw
    static bool writeZip(const std::string &path, const float *rgb, int width, int height, bool half, const std::string &log, int compression)
    {
        std::vector<char> hdr;
        header(hdr, width, height, half, log, compression);
        const int lines = compression == ZIP_COMPRESSION ? 16 : 1, chunks = (height + lines - 1) / lines;
        const size_t bpc = half ? 2 : 4, lineBytes = (size_t)width * 3 * bpc;
        std::vector<std::vector<char>> blobs(chunks);
        std::vector<unsigned char> raw, tmp;
        for (int cidx = 0; cidx < chunks; ++cidx) {
            const int y0 = cidx * lines, y1 = std::min(height, y0 + lines);
            raw.assign((size_t)(y1 - y0) * lineBytes, 0);
            for (int y = y0; y < y1; ++y) packLine(&raw[(size_t)(y - y0) * lineBytes], rgb, y, width, half);
            tmp.resize(raw.size());
            unsigned char *t1 = tmp.data(), *t2 = tmp.data() + (raw.size() + 1) / 2;
            for (size_t i = 0; i < raw.size(); ++i) { if (i & 1) *t2++ = raw[i]; else *t1++ = raw[i]; }
            { int p = tmp[0]; for (size_t i = 1; i < tmp.size(); ++i) { const int d = (int)tmp[i] - p + (128 + 256); p = tmp[i]; tmp[i] = (unsigned char)d; } }
            uLongf outLen = compressBound((uLong)tmp.size());
            std::vector<char> z(outLen);
            if (compress(reinterpret_cast<Bytef *>(z.data()), &outLen, tmp.data(), (uLong)tmp.size()) != Z_OK) return false;
            if (outLen >= raw.size()) blobs[cidx].assign(raw.begin(), raw.end());
            else { z.resize(outLen); blobs[cidx].swap(z); }
        }
        std::ofstream f(path, std::ios::binary);
        if (!f) return false;
        f.write(hdr.data(), hdr.size());
        uint64_t off = hdr.size() + (uint64_t)chunks * 8;
        for (int cidx = 0; cidx < chunks; ++cidx) { f.write(reinterpret_cast<const char *>(&off), 8); off += 8 + blobs[cidx].size(); }
        for (int cidx = 0; cidx < chunks; ++cidx) {
            const int32_t yy = cidx * lines, sz = (int32_t)blobs[cidx].size();
            f.write(reinterpret_cast<const char *>(&yy), 4);
            f.write(reinterpret_cast<const char *>(&sz), 4);
            f.write(blobs[cidx].data(), blobs[cidx].size());
        }
        return (bool)f;
    }

    /// Radiance RGBE (`fileFormat=rgbe`): shared-exponent bytes as Bitmap::writeRGBE converts them (src/libcore/bitmap.cpp: max component -> frexp)
    static bool writeRGBE(const std::string &path, const float *rgb, int width, int height)
    {
        std::ofstream f(path, std::ios::binary);
        if (!f) return false;
        f << "#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y " << height << " +X " << width << "\n";
        std::vector<unsigned char> line((size_t)width * 4);
        for (int y = 0; y < height; ++y) {
            for (int x = 0; x < width; ++x) {
                const float *v = rgb + ((size_t)y * width + x) * 3;
                const float m = std::max(v[0], std::max(v[1], v[2]));
                unsigned char *o = &line[(size_t)x * 4];
                if (!(m >= 1e-32f)) { o[0] = o[1] = o[2] = o[3] = 0; continue; }
                int e;
                const float sc = std::frexp(m, &e) * 256.0f / m;
                o[0] = (unsigned char)(v[0] > 0 ? v[0] * sc : 0); o[1] = (unsigned char)(v[1] > 0 ? v[1] * sc : 0); o[2] = (unsigned char)(v[2] > 0 ? v[2] * sc : 0);
                o[3] = (unsigned char)(e + 128);
            }
            f.write(reinterpret_cast<const char *>(line.data()), line.size());
        }
        return (bool)f;
    }

private:
    static void packLine(unsigned char *line, const float *rgb, int y, int width, bool half)
    {
        for (int c = 0; c < 3; ++c) {                               // file order B, G, R  <-  memory order R, G, B
            const int src = 2 - c;
            for (int x = 0; x < width; ++x) {
                const float v = rgb[((size_t)y * width + x) * 3 + src];
                if (half) { const uint16_t h = float_to_half(v); std::memcpy(&line[((size_t)c * width + x) * 2], &h, 2); }
                else std::memcpy(&line[((size_t)c * width + x) * 4], &v, 4);
            }
        }
    }
    static void header(std::vector<char> &hdr, int width, int height, bool half, const std::string &log, int compression)
    {
        put32(hdr, 20000630);       // magic
        put32(hdr, 2);              // version 2, single-part scanline
        {
            std::vector<char> ch;
            for (const char *nm : {"B", "G", "R"}) {
                ch.push_back(nm[0]); ch.push_back(0);
                put32(ch, half ? 1 : 2);
                ch.push_back(0); ch.push_back(0); ch.push_back(0); ch.push_back(0);
                put32(ch, 1); put32(ch, 1);
            }
            ch.push_back(0);
            attr(hdr, "channels", "chlist", ch);
        }
        attr(hdr, "compression", "compression", std::vector<char>(1, (char)compression));
        { std::vector<char> b; put32(b, 0); put32(b, 0); put32(b, width - 1); put32(b, height - 1); attr(hdr, "dataWindow", "box2i", b); attr(hdr, "displayWindow", "box2i", b); }
        { const std::string g = "gdpt-mi355x (gradient-domain path tracer, HIP)"; attr(hdr, "generatedBy", "string", std::vector<char>(g.begin(), g.end())); }
        attr(hdr, "lineOrder", "lineOrder", std::vector<char>(1, 0));
        if (!log.empty()) attr(hdr, "log", "string", std::vector<char>(log.begin(), log.end()));     // multifilm.cpp:478-506 attaches the log
        { std::vector<char> f; putf(f, 1.0f); attr(hdr, "pixelAspectRatio", "float", f); }
        { std::vector<char> v; putf(v, 0.0f); putf(v, 0.0f); attr(hdr, "screenWindowCenter", "v2f", v); }
        { std::vector<char> f; putf(f, 1.0f); attr(hdr, "screenWindowWidth", "float", f); }
        hdr.push_back(0);
    }
    static void put32(std::vector<char> &b, int32_t v) { const char *p = reinterpret_cast<const char *>(&v); b.insert(b.end(), p, p + 4); }
    static void putf(std::vector<char> &b, float v) { const char *p = reinterpret_cast<const char *>(&v); b.insert(b.end(), p, p + 4); }
    static void attr(std::vector<char> &b, const char *name, const char *type, const std::vector<char> &data)
    {
        b.insert(b.end(), name, name + std::strlen(name) + 1);
        b.insert(b.end(), type, type + std::strlen(type) + 1);
        put32(b, (int32_t)data.size());
        b.insert(b.end(), data.begin(), data.end());
    }
};

} // namespace gdpt
