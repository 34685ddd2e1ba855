// png.cpp -- PNG (8-bit RGB, zlib deflate) and base64 encoders for the host
// twins of app/Write.hs and app/Web.hs.  png.h is not in the image, so the
// container format is written by hand over zlib.  The PIXELS are what the
// reference's repa3ToImage defines; the PNG byte stream itself is not claimed
// identical to JuicyPixels' encoder (parity unpinned by the reference).
#include "pathtracer.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstring>

namespace pathtracer {
namespace Conversion {

namespace {
void put32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
void chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &data)
{
    put32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
}
} // namespace

std::vector<uint8_t> encodePng(const Image &img)
{
    if (img.width < 1 || img.height < 1 || img.rgb.size() != (size_t)img.width * img.height * 3)
        throw std::invalid_argument("encodePng: image size mismatch");
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)img.width); put32(ihdr, (uint32_t)img.height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    // filter type 0 in front of every scanline
    const size_t row = (size_t)img.width * 3;
    std::vector<uint8_t> raw((row + 1) * img.height);
    for (int y = 0; y < img.height; ++y) {
        raw[(row + 1) * y] = 0;
        std::memcpy(&raw[(row + 1) * y + 1], &img.rgb[row * y], row);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK)
        throw std::runtime_error("encodePng: zlib compress2 failed");
    z.resize(zlen);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    return out;
}

std::string base64(const std::vector<uint8_t> &b)
{
    static const char tbl[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string s;
    s.reserve((b.size() + 2) / 3 * 4);
    size_t i = 0;
    for (; i + 2 < b.size(); i += 3) {
        const uint32_t v = (b[i] << 16) | (b[i + 1] << 8) | b[i + 2];
        s += tbl[v >> 18]; s += tbl[(v >> 12) & 63]; s += tbl[(v >> 6) & 63]; s += tbl[v & 63];
    }
    if (i + 1 == b.size()) {
        const uint32_t v = b[i] << 16;
        s += tbl[v >> 18]; s += tbl[(v >> 12) & 63]; s += "==";
    } else if (i + 2 == b.size()) {
        const uint32_t v = (b[i] << 16) | (b[i + 1] << 8);
        s += tbl[v >> 18]; s += tbl[(v >> 12) & 63]; s += tbl[(v >> 6) & 63]; s += '=';
    }
    return s;
}

void savePngImage(const std::string &path, const Image &img)
{
    const std::vector<uint8_t> png = encodePng(img);
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("savePngImage: cannot open " + path);
    const size_t n = std::fwrite(png.data(), 1, png.size(), f);
    std::fclose(f);
    if (n != png.size()) throw std::runtime_error("savePngImage: short write to " + path);
}

} // namespace Conversion
} // namespace pathtracer
