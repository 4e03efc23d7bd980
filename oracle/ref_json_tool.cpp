// ORACLE / test infrastructure: the ONE piece of the reference that compiles here on its own -- its vendored JSON library
// include/OpenCameraCalibrator/utils/json.h (nlohmann::json 3.7.0), which the reference uses for every file it reads and
// writes (src/io/read_scene.cc:25-41 from_ubjson; src/core/board_extractor.cc:245-266 to_ubjson; result / calibration
// JSON with std::setw).  Built by oracle/Makefile (target `ref`) FROM THE SOURCES WHERE THEY LIE under /root/reference
// (-I/root/reference/include; nothing is copied) into oracle/_ref/ref_json_tool.  tests/golden/make_ubjson_golden.py uses
// it to produce the committed fixtures that pin this repository's UBJSON / JSON readers and writers to the reference's
// own serializer.
//   ref_json_tool to_ubjson   in.json  out.uson     nlohmann::json::to_ubjson(parse(in))
//   ref_json_tool from_ubjson in.uson  out.json     from_ubjson, dumped with std::setw(4) like the reference's writers
#include <fstream>
#include <iomanip>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "OpenCameraCalibrator/utils/json.h"

int main(int argc, char** argv) {
  if (argc != 4) { std::cerr << "usage: ref_json_tool to_ubjson|from_ubjson IN OUT\n"; return 2; }
  const std::string mode = argv[1];
  if (mode == "to_ubjson") {
    std::ifstream in(argv[2]); nlohmann::json j; in >> j;
    const std::vector<std::uint8_t> b = nlohmann::json::to_ubjson(j);
    std::ofstream out(argv[3], std::ios::binary); out.write(reinterpret_cast<const char*>(b.data()), std::streamsize(b.size()));
    return 0;
  }
  if (mode == "from_ubjson") {
    std::ifstream in(argv[2], std::ios::binary);
    const std::vector<std::uint8_t> b((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    const nlohmann::json j = nlohmann::json::from_ubjson(b);
    std::ofstream out(argv[3]); out << std::setw(4) << j << std::endl;
    return 0;
  }
  return 2;
}
