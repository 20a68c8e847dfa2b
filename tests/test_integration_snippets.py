"""The C++ of INTEGRATION.md sections 2 and 3 -- what a maintainer pastes into the reference's include/cvo.hpp and src/cvo.cpp
(ref src/cvo.cpp:18-48,319-420, src/adaptive_cvo.cpp:385-439) -- in front of a compiler, against include/cvo_hip.h and ten-line
stand-ins for Eigen and the reference's cloud types (tests/cpp/integration_stubs.hpp).  A boundary TYPO CHECK (names, argument
counts, pointer types of the C-ABI calls): it pins nothing about results and is no parity evidence."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpp_blocks(section_from, section_to):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    part = text[text.index(section_from):text.index(section_to)]
    return re.findall(r"```cpp\n(.*?)```", part, re.S)


def no_ellipsis(block):
    out = []
    for ln in block.splitlines():
        if ln.strip() == "..." or ln.strip().startswith("... //"):
            continue
        out.append(ln.replace("init(false), ... {", "init(false) {"))
    return "\n".join(out)


def test_integration_snippets_compile(tmp_path):
    hpp = cpp_blocks("## 2. Edits", "## 3. Edits")
    src = cpp_blocks("## 3. Edits", "## 4. Error")
    assert len(hpp) == 1 and len(src) == 5, (len(hpp), len(src))
    members = "\n".join(ln for ln in no_ellipsis(hpp[0]).splitlines() if "hip_" in ln and "#include" not in ln)
    ctor, set_pcd, align, four_ops, fip = [no_ellipsis(b) for b in src]
    code = """
#include "cvo_hip.h"
#include "integration_stubs.hpp"
namespace cvo {
class cvo {
  public:
    cvo(); ~cvo();
    void set_pcd(const cv::Mat &RGB_img, const cv::Mat &dep_img);
    void align();
    void loop_body();
    float function_inner_product(point_cloud *cloud_a, point_cloud *cloud_b);
  private:
    bool init; int iter = 0; float ell = 0.15f, min_step = 0.2f, step = 0.f;
    pcd_generator pcd_gen;
    std::unique_ptr<frame> ptr_fixed_fr, ptr_moving_fr;
    std::unique_ptr<point_cloud> ptr_fixed_pcd, ptr_moving_pcd;
    Eigen::Matrix3f R; Eigen::Vector3f T, omega, v;
    Eigen::Affine3f transform, prev_transform, accum_transform;
%s
};
%s
void cvo::set_pcd(const cv::Mat &RGB_img, const cv::Mat &dep_img) {
%s
}
%s
void cvo::loop_body() {
%s
}
float cvo::function_inner_product(point_cloud *cloud_a, point_cloud *cloud_b) {
%s
}
}   // namespace cvo
int main() { return 0; }
""" % (members, ctor, set_pcd, align, four_ops, fip)
    f = tmp_path / "integration_snippets.cpp"
    f.write_text(code)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp"), str(f)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
