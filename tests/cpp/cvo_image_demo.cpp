// cvo_image_demo.cpp -- the reference's driver loop (ref src/cvo_main.cpp:17-66,
// adaptive_cvo_main.cpp) on the C++ mirror objects with IMAGES as input: every
// frame goes through run_cvo(dataset_seq, RGB_img, dep_img), which runs the GPU
// front end and the registration; a pose line per frame goes to stdout in the
// reference's format (default ostream precision).
// Input: a binary file written by the test: int32 n_frames, width, height; per
// frame 32 bytes of name (zero padded), height*width*3 bytes (B,G,R as cv::imread
// decodes) and height*width uint16 depth.  (PNG decoding is the caller's: the
// reference uses cv::imread, ref src/cvo_main.cpp:100-106.)
// Build: g++ -std=c++17 -I include cvo_image_demo.cpp -L cvo-rgbd_amd/csrc -lcvo_hip
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "cvo.hpp"

int main(int argc, char **argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: demo frames.bin cvo|acvo dataset_seq\n"); return 2; }
    std::ifstream in(argv[1], std::ios::binary);
    int32_t hdr[3] = {0, 0, 0};
    in.read(reinterpret_cast<char *>(hdr), 12);
    const int nf = hdr[0], w = hdr[1], h = hdr[2];
    const bool adaptive = std::string(argv[2]) == "acvo";
    const int dataset_seq = std::stoi(argv[3]);
    std::ostringstream poses, quiet;
    std::streambuf *old = std::cout.rdbuf(quiet.rdbuf());   // run_cvo prints like the reference does
    try {
        std::unique_ptr<cvo_hip::registration> reg;
        if (adaptive) reg.reset(new acvo::acvo());
        else reg.reset(new cvo::cvo());
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        std::vector<uint16_t> dep((size_t)w * h);
        for (int i = 0; i < nf; ++i) {
            char name[33] = {0};
            in.read(name, 32);
            in.read(reinterpret_cast<char *>(rgb.data()), (std::streamsize)rgb.size());
            in.read(reinterpret_cast<char *>(dep.data()), (std::streamsize)dep.size() * 2);
            if (!in) { std::fprintf(stderr, "short read\n"); return 2; }
            const cvo_hip::image_view RGB_img{rgb.data(), h, w, (size_t)w * 3};
            const cvo_hip::image_view dep_img{dep.data(), h, w, (size_t)w * 2};
            reg->run_cvo(dataset_seq, RGB_img, dep_img, "unused.pcd", "unused.pcd");
            if (reg->init) {   // ref src/cvo_main.cpp:58-65
                float q[4];
                reg->accum_transform.quaternion(q);
                poses << name << " ";
                poses << reg->accum_transform.matrix()(0, 3) << " " << reg->accum_transform.matrix()(1, 3) << " "
                      << reg->accum_transform.matrix()(2, 3) << " ";
                poses << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << "\n";
            }
        }
        std::cout.rdbuf(old);
        std::cout << poses.str();
        std::cout << "points_last_frame " << reg->num_points_last_frame() << " iterations " << reg->num_iterations()
                  << "\n";
    } catch (const std::exception &e) {
        std::cout.rdbuf(old);
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
