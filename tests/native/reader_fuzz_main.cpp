// Test infrastructure: runs the model readers (TFLite flatbuffer, ONNX protobuf), the operand validation and the graph
// passes over a corpus of mutated model files under AddressSanitizer + UBSan (built by tests/test_readers_asan.py with g++;
// no HIP needed).  Corpus file format: repeated { u32 length, bytes }.  Exit code 0 = every blob was either accepted or
// rejected with an error; any out-of-bounds access aborts the process with a sanitizer report.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../birdnet-go_amd/csrc/model_onnx.h"
#include "../../birdnet-go_amd/csrc/tflite_model.h"

using namespace bnhip;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> all((std::istreambuf_iterator<char>(f)), {});
    size_t pos = 0, n_blobs = 0, accepted = 0;
    while (pos + 4 <= all.size()) {
        uint32_t len; memcpy(&len, &all[pos], 4); pos += 4;
        if (len > all.size() - pos) return 3;
        // exact-size heap copy so that a read past the end is a heap-buffer-overflow, not a read of the next blob
        std::vector<char> blob(all.begin() + pos, all.begin() + pos + len);
        pos += len; n_blobs++;
        TflModel m; std::string err; int code = 0;
        bool ok;
        if (blob.size() >= 8 && memcmp(blob.data() + 4, "TFL3", 4) == 0) ok = parse_tflite(blob.data(), blob.size(), &m, &err);
        else ok = parse_onnx(blob.data(), blob.size(), &m, &err, &code);
        if (ok) ok = validate_graph(m, &err);
        if (ok) ok = run_graph_passes(&m, &err);
        if (ok) ok = validate_graph(m, &err);              // the passes must leave a valid graph behind
        accepted += ok;
    }
    printf("blobs %zu accepted %zu\n", n_blobs, accepted);
    return 0;
}
