// selftest.cpp -- exercises the host codecs (PNG, base64) without a GPU:
// writes a deterministic W x H test image to argv[1] and prints its
// imageToText on stdout.  tests/test_host_cpu.py decodes both.
#include "pathtracer.hpp"

#include <cstdio>
#include <cstdlib>

using namespace pathtracer;

int main(int argc, char **argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: selftest out.png W H\n"); return 2; }
    Conversion::Image img{std::atoi(argv[2]), std::atoi(argv[3]), {}};
    img.rgb.resize((size_t)img.width * img.height * 3);
    for (int y = 0; y < img.height; ++y)
        for (int x = 0; x < img.width; ++x)
            for (int k = 0; k < 3; ++k)
                img.rgb[((size_t)y * img.width + x) * 3 + k] = (uint8_t)((x * 7 + y * 13 + k * 101) & 255);
    Conversion::savePngImage(argv[1], img);
    std::printf("%s\n", Conversion::imageToText(img).c_str());
    return 0;
}
