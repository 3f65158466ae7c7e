"""Generates the committed fixtures from the reference's own test data (runs only where /root/reference exists).

  test_images.npz  - tests/test_images/{1,2}.png decoded to BGR uint8 (what cv::imread(IMREAD_COLOR) returns)
  utils_points.npz - the 79 point pairs and the golden F / H of src/tests/geometry/utils_tests.cpp:15-121
"""
import os
import re

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    imgs = {}
    for i in (1, 2):
        rgb = np.array(Image.open(f"{REF}/tests/test_images/{i}.png").convert("RGB"))
        imgs[f"bgr{i}"] = np.ascontiguousarray(rgb[:, :, ::-1])
    np.savez_compressed(os.path.join(HERE, "test_images.npz"), **imgs)

    src = open(f"{REF}/src/tests/geometry/utils_tests.cpp").read()

    def grab(name):
        i = src.index("std::vector<cv::Point2f> %s =" % name)
        j = src.index("};", i)
        body = src[src.index("{", i):j]
        v = [float(x) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", body)]
        return np.array(v, np.float32).reshape(-1, 2)

    def grab_mat(tag):
        i = src.index(tag)
        i = src.index("<<", i)
        j = src.index(");", i)
        body = re.sub(r"//[^\n]*", "", src[i:j])
        v = [float(x) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", body)]
        assert len(v) == 9
        return np.array(v, np.float64).reshape(3, 3)

    np.savez(os.path.join(HERE, "utils_points.npz"), ref_points=grab("ref_points"), points=grab("points"),
             F_gold=grab_mat("cv_f_mat =\n"), H_gold=grab_mat("cv_h_mat =\n"))


if __name__ == "__main__":
    main()
