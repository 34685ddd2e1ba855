// frames.cpp -- the pull-next-frame contract of app/Web.hs:42-65 without the
// web server (Yesod/WebSocket transport is out of scope):
//     imageSource = sourceList traces ; formatAsImgSrc = ("data:image/png;base64," <>) . repa3ToText
// Every line read from stdin stands for one client message and is answered
// with the next element of `traces` as a data URI on stdout.  Frame k costs
// ONE more traceCanvas on the GPU (the reference recomputes all k, O(k^2) per
// session); the first frame is the black canvas (traces !! 0).
#include "pathtracer.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

using namespace pathtracer;

int main(int argc, char **argv)
{
    Params p;
    int mode = PT_RGB8_REFERENCE;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!std::strcmp(argv[i], "--height")) p.height = std::atoi(argv[i + 1]);
        else if (!std::strcmp(argv[i], "--width")) p.width = std::atoi(argv[i + 1]);
        else if (!std::strcmp(argv[i], "--bounces")) p.maxBounces = std::atoi(argv[i + 1]);
        else if (!std::strcmp(argv[i], "--mode")) mode = !std::strcmp(argv[i + 1], "mean") ? PT_RGB8_MEAN_CLAMP : PT_RGB8_REFERENCE;
    }
    try {
        Lib::Traces traces(p, objects());
        std::string line;
        int64_t k = 0;
        while (std::getline(std::cin, line)) {
            traces.advance(k++);
            std::cout << "data:image/png;base64," << Conversion::imageToText(Conversion::repa3ToImage(traces, mode))
                      << std::endl;
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "frames: %s\n", e.what());
        return 1;
    }
    return 0;
}
