"""Writes tests/golden/ref_corners.uson / ref_corners.json / ref_roundtrip.json with the REFERENCE's own JSON library
(oracle/_ref/ref_json_tool = include/OpenCameraCalibrator/utils/json.h of /root/reference, built by `make -C oracle ref`):
  ref_corners.json   a corner file as text (what extract_board_to_json holds before serialising; board_extractor.cc:245-266)
  ref_corners.uson   nlohmann::json::to_ubjson of it -- the bytes the reference writes and src/io/read_scene.cc reads
  ref_roundtrip.json from_ubjson of those bytes dumped with std::setw(4) (the reference's output style)
The fixtures pin this repository's UBJSON encoder / decoders (Python io_files, C++ host/json_min.hpp) to the reference's
serializer.  Needs /root/reference (build container only); the tests read only the committed files.
    python tests/golden/make_ubjson_golden.py"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from openimucameracalibrator_amd import camera_calibrator as CC  # noqa: E402


def scene():
    ds = CC.make_calibration_dataset("gopro9_division", num_views=12, corners_per_view=40, seed=11)
    views = {}
    for v in range(len(ds["pose_true"])):
        a, b = ds["corner_offset"][v], ds["corner_offset"][v + 1]
        views[str(1600000000000000 + 33333 * v)] = dict(
            image_points={str(int(ds["point_ids"][c])): [float(ds["uv"][c, 0]), float(ds["uv"][c, 1])] for c in range(a, b)})
    # integer widths of every UBJSON integer marker, negative numbers, a float that is integral, bool, null, string
    extras = dict(ints=[0, 5, -5, 127, 128, 255, 256, -129, 32767, 32768, -32769, 2147483647, 2147483648, -2147483649, 1600000000000000000],
                  floats=[0.5, -1.25e-7, 3.0e300, 1.0], flag=True, nothing=None, text="charuco 9x7")
    return dict(views=views, scene_pts={str(i): ds["points"][i, :3].tolist() for i in range(len(ds["points"]))},
                image_width=ds["width"], image_height=ds["height"], camera_fps=59.94, extras=extras)


if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "-s"])
    tool = os.path.join(ROOT, "oracle", "_ref", "ref_json_tool")
    assert os.path.exists(tool), "reference not mounted: cannot regenerate"
    p = lambda n: os.path.join(HERE, n)
    json.dump(scene(), open(p("ref_corners.json"), "w"))
    subprocess.check_call([tool, "to_ubjson", p("ref_corners.json"), p("ref_corners.uson")])
    subprocess.check_call([tool, "from_ubjson", p("ref_corners.uson"), p("ref_roundtrip.json")])
    print({n: os.path.getsize(p(n)) for n in ("ref_corners.json", "ref_corners.uson", "ref_roundtrip.json")})
