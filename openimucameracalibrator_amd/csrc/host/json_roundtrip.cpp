// Test utility (not a product application): parse a UBJSON or JSON file with host/json_min.hpp and write it back as JSON with a
// given indent.  tests/test_ref_json_fixture.py compares the result with what the reference's own library (nlohmann::json
// 3.7.0, include/OpenCameraCalibrator/utils/json.h) prints for the same bytes.
//   json_roundtrip IN OUT INDENT
#include <fstream>
#include <iostream>

#include "json_min.hpp"

int main(int argc, char** argv) {
  if (argc != 4) { std::cerr << "usage: json_roundtrip IN OUT INDENT\n"; return 2; }
  std::string bytes;
  if (!oicc_json::read_file(argv[1], &bytes)) { std::cerr << "cannot read " << argv[1] << "\n"; return 1; }
  const std::string path = argv[1];
  const bool text = path.size() > 5 && path.substr(path.size() - 5) == ".json";
  try {
    const oicc_json::Value v = text ? oicc_json::Parser::parse(bytes) : oicc_json::UbjsonReader::parse(bytes);
    std::ofstream out(argv[2]);
    oicc_json::dump(v, out, std::stoi(argv[3])); out << std::endl;
  } catch (const std::exception& e) { std::cerr << e.what() << "\n"; return 1; }
  return 0;
}
