// Minimal, bounds-checked ONNX (protobuf) reader for the float graphs the reference loads through ONNX Runtime:
// embedding -> class heads (CustomClassifier, internal/inference/onnx/custom_classifier.go:148-174; BattyBirdNET regional
// heads, internal/classifier/bat_onnx.go:252-282) and range-filter style dense stacks.  The graph is lowered onto the same
// operator IR the TFLite reader produces (TflModel), so one planner serves both containers.
#pragma once
#include <cstddef>
#include <string>

#include "tflite_model.h"

namespace bnhip {

// false + err + *code (BNHIP_E_MODEL for a malformed file, BNHIP_E_UNSUPPORTED for a well-formed graph using operators
// or features outside the supported set).  Never reads out of bounds.
bool parse_onnx(const void* blob, size_t n, TflModel* out, std::string* err, int* code);

}  // namespace bnhip
