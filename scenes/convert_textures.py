#!/usr/bin/env python3
"""Pre-decodes the image files a scene names into binary PNM next to them ("<file>.ppm" colour, "<file>.pgm" grey).

The reference decodes images with cv::imread (OpenCV, not available to this build); the host library and the oracle
read binary PNM only and look for these siblings.  Run where PIL is available:

    python scenes/convert_textures.py SRC_DIR DST_DIR file.jpg[:grey] ...

`:grey` asks the JPEG decoder for its luma channel directly, which is what cv::imread(..., GRAYSCALE) does.
Texel values of lossy formats depend on the decoder: parity of texels with the reference is unpinned.
"""
import os
import sys
from PIL import Image


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for spec in sys.argv[3:]:
        name, _, mode = spec.partition(":")
        im = Image.open(os.path.join(src, name))
        if mode == "grey":
            im.draft("L", im.size)
            im = im.convert("L")
            out = os.path.join(dst, name + ".pgm")
            header = b"P5\n%d %d\n255\n" % im.size
        else:
            im = im.convert("RGB")
            out = os.path.join(dst, name + ".ppm")
            header = b"P6\n%d %d\n255\n" % im.size
        with open(out, "wb") as f:
            f.write(header + im.tobytes())
        print(out, im.size)


if __name__ == "__main__":
    main()
