// calibdata_to_json -- export a TheiaSfM pose data set (.calibdata, the cereal binary that the reference's
// estimate_camera_poses_from_checkerboard writes, applications/estimate_camera_poses_from_checkerboard.cc:71, and its
// continuous_time_imu_to_camera_calibration reads, applications/continuous_time_imu_to_camera_calibration.cc:95-97) to the JSON
// twin that this repository's applications read (csrc/host/cli_common.hpp read_pose_dataset, openimucameracalibrator_amd/io_files.py):
//
//   { "views":  { "<view name = image time in us>": { "orientation_angle_axis": [rx, ry, rz],   // theia::Camera::GetOrientationAsAngleAxis (world -> camera)
//                                                      "position": [x, y, z] }, ... },             // theia::Camera::GetPosition (camera centre in the world)
//     "tracks": { "<track id = board point id>": [x, y, z, w], ... } }                             // theia::Track::Point (homogeneous)
//
// It has to run where TheiaSfM exists (the reference's own build environment, Dockerfile:40-44): the archive format is TheiaSfM's.
// It uses nothing but the calls the reference itself makes on these objects (continuous_time_imu_to_camera_calibration.cc:107-150).
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (no TheiaSfM / Eigen here); tests/test_pose_twin_schema.py holds the schema it must produce.
//
//   g++ -std=c++17 calibdata_to_json.cc -o calibdata_to_json $(pkg-config --cflags eigen3) -ltheia -lglog -lgflags   (as the reference links TheiaSfM)
//   calibdata_to_json pose_calib.calibdata pose_calib.json
#include <theia/theia.h>

#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>

int main(int argc, char* argv[]) {
  if (argc != 3) { std::cerr << "usage: calibdata_to_json <input.calibdata> <output.json>\n"; return 2; }
  theia::Reconstruction pose_dataset;
  if (!theia::ReadReconstruction(argv[1], &pose_dataset)) { std::cerr << "Could not read Reconstruction file " << argv[1] << "\n"; return 1; }
  std::ofstream out(argv[2]);
  if (!out) { std::cerr << "Could not open " << argv[2] << "\n"; return 1; }
  out << std::setprecision(17) << "{\n  \"views\": {";
  bool first = true;
  for (const theia::ViewId view_id : pose_dataset.ViewIds()) {
    const theia::View* view = pose_dataset.View(view_id);
    if (view == nullptr) continue;                                              // (every view: the calibration looks them up by name, :138-142)
    const theia::Camera& cam = view->Camera();
    const Eigen::Vector3d aa = cam.GetOrientationAsAngleAxis(), c = cam.GetPosition();
    out << (first ? "\n" : ",\n") << "    \"" << view->Name() << "\": {\"orientation_angle_axis\": [" << aa[0] << ", " << aa[1] << ", " << aa[2]
        << "], \"position\": [" << c[0] << ", " << c[1] << ", " << c[2] << "]}";
    first = false;
  }
  out << "\n  },\n  \"tracks\": {";
  first = true;
  for (const theia::TrackId track_id : pose_dataset.TrackIds()) {
    const Eigen::Vector4d& X = pose_dataset.Track(track_id)->Point();
    out << (first ? "\n" : ",\n") << "    \"" << track_id << "\": [" << X[0] << ", " << X[1] << ", " << X[2] << ", " << X[3] << "]";
    first = false;
  }
  out << "\n  }\n}\n";
  std::cout << "wrote " << pose_dataset.NumViews() << " views, " << pose_dataset.NumTracks() << " tracks to " << argv[2] << "\n";
  return 0;
}
