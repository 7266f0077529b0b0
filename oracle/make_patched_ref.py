#!/usr/bin/env python3
"""Build the reference WITH the drop-in patch of INTEGRATION.md section 2 applied (test
infrastructure): the reference's own marker parser and Huffman decoder in front of libjpegblk.so.

  python oracle/make_patched_ref.py [/root/reference]   ->  oracle/_ref/jpeg_patched

The patch text is taken from INTEGRATION.md itself, so what is tested is what is documented.  The
patched source is a build intermediate in a temporary directory (reference sources are never
copied into the repo); only the binary lands in oracle/_ref/ (git-ignored, travels with gpurun).
For the test a single extra line is appended to the patch: when JB_DUMP names a file, the decoded
picture is written there as PPM (the reference's only sink is an X11 window).
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    src_path = os.path.join(ref, "jpeg.cpp")
    if not os.path.exists(src_path):
        print(f"reference not present at {ref}: keeping prebuilt oracle/_ref/jpeg_patched (if any)")
        return 0
    src = open(src_path).read()
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    a = md.index("  {\n    jb_image_desc d{")
    end = "    jb_pinned_free(coef);\n  }\n"
    block = md[a:md.index(end, a) + len(end)]
    dump = ('    if (getenv("JB_DUMP")) jb_write_ppm(getenv("JB_DUMP"), rgb.data(), image_width, image_height, 3LL * image_width);\n'
            "    jb_pinned_free(coef);\n  }\n")
    block = block.replace(end, dump)
    seam = "    dequantize();\n    inverseDCT();\n    YCbCrToRGB();\n"      # jpeg.cpp:786-788
    if seam not in src:
        print("the seam (jpeg.cpp:786-788) was not found verbatim", file=sys.stderr)
        return 1
    src = src.replace(seam, block)
    m = re.search(r"class Image\s*\{", src)
    src = src[:m.end()] + "\n  jb_ctx *jb = nullptr;\n  std::vector<uint8_t> rgb;\n" + src[m.end():]
    src = '#include "jpegblk.h"\n#include <vector>\n#include <cstdint>\n#include <cstdlib>\n' + src
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    lib_dir = os.path.join(ROOT, "jpeg_decoder_amd")
    with tempfile.TemporaryDirectory() as tmp:
        cpp = os.path.join(tmp, "jpeg_patched.cpp")
        with open(cpp, "w") as f:
            f.write(src)
        cmd = ["g++", "-std=c++17", "-O1", "-w", "-U_FORTIFY_SOURCE", "-D_FORTIFY_SOURCE=0",
               "-I" + os.path.join(ROOT, "include"), "-I" + ref, "-I" + os.path.join(ref, "include"), cpp,
               "-L" + lib_dir, "-ljpegblk", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,$ORIGIN/../../jpeg_decoder_amd",
               "-lX11", "-o", os.path.join(out_dir, "jpeg_patched")]
        subprocess.run(cmd, check=True)
    print("built oracle/_ref/jpeg_patched (reference + INTEGRATION.md patch)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
