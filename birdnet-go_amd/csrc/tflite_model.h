// Minimal, bounds-checked reader for TFLite flatbuffers (float models).
//
// The reference passes its classifier backend the model as an in-memory byte slice
// (internal/classifier/birdnet.go:1195-1246 -> internal/inference/tflite/classifier.go:38-41), so the
// container is part of the drop-in boundary.  Schema: TensorFlow Lite 2.17.1 schema.fbs (third-party).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace bnhip {

enum TensorType : int { TT_FLOAT32 = 0, TT_FLOAT16 = 1, TT_INT32 = 2, TT_UINT8 = 3, TT_INT64 = 4,
                        TT_COMPLEX64 = 8, TT_INT8 = 9 };

// BuiltinOperator codes (subset; numbering of TensorFlow Lite 2.17.1 schema.fbs)
enum OpCode : int {
    OP_NOP = -2,             // removed by a graph pass (graph_passes.cpp)
    OP_ADD = 0, OP_AVERAGE_POOL_2D = 1, OP_CONCATENATION = 2, OP_CONV_2D = 3, OP_DEPTHWISE_CONV_2D = 4,
    OP_DEQUANTIZE = 6, OP_FLOOR = 8, OP_FULLY_CONNECTED = 9, OP_LOGISTIC = 14, OP_MAX_POOL_2D = 17, OP_MUL = 18, OP_RELU = 19,
    OP_RELU_N1_TO_1 = 20, OP_RELU6 = 21, OP_RESHAPE = 22, OP_SOFTMAX = 25, OP_TANH = 28, OP_PAD = 34, OP_GATHER = 36,
    OP_TRANSPOSE = 39, OP_MEAN = 40, OP_SUB = 41, OP_DIV = 42, OP_SQUEEZE = 43, OP_STRIDED_SLICE = 45, OP_EXP = 47,
    OP_SPLIT = 49, OP_CAST = 53, OP_MAXIMUM = 55, OP_MINIMUM = 57, OP_NEG = 59, OP_PADV2 = 60, OP_SLICE = 65, OP_SIN = 66,
    OP_EXPAND_DIMS = 70, OP_LOG = 73, OP_SUM = 74, OP_SQRT = 75, OP_RSQRT = 76, OP_POW = 78, OP_REDUCE_PROD = 81,
    OP_REDUCE_MAX = 82, OP_REDUCE_MIN = 89, OP_SQUARE = 92, OP_LEAKY_RELU = 98, OP_SQUARED_DIFFERENCE = 99, OP_ABS = 101,
    OP_CEIL = 104, OP_REVERSE_V2 = 105, OP_COS = 108, OP_ELU = 111, OP_ROUND = 116, OP_HARD_SWISH = 117,
    OP_BATCH_MATMUL = 126, OP_RFFT2D = 131, OP_IMAG = 133, OP_REAL = 134, OP_COMPLEX_ABS = 135, OP_GELU = 150
};
const char* op_name(int code);

struct TflTensor {
    std::string name;
    std::vector<int> shape;
    int type = 0;
    const uint8_t* data = nullptr;   // points into the caller's blob (valid during model_create only)
    size_t nbytes = 0;
    size_t numel() const { size_t n = 1; for (int d : shape) n *= (size_t)d; return n; }
    const float* f32() const { return reinterpret_cast<const float*>(data); }
    const int32_t* i32() const { return reinterpret_cast<const int32_t*>(data); }
};

struct TflOp {
    int code = -1;
    std::vector<int> inputs, outputs;
    // decoded option fields (only those relevant to `code` are meaningful)
    int padding = 0, stride_w = 1, stride_h = 1, dil_w = 1, dil_h = 1, act = 0, depth_multiplier = 1;
    int filter_w = 0, filter_h = 0;
    int axis = 0, batch_dims = 0;
    bool keep_dims = false, keep_num_dims = false, adj_x = false, adj_y = false;
    int in_type = 0, out_type = 0;
    float beta = 1.0f, alpha = 0.0f;
    int begin_mask = 0, end_mask = 0, ellipsis_mask = 0, new_axis_mask = 0, shrink_axis_mask = 0, num_splits = 0;
    bool approximate = false;          // GELU
    std::vector<int> new_shape, squeeze_dims;
    // explicit zero padding folded in from a preceding PAD by graph_passes.cpp (then `padding` is ignored)
    bool explicit_pad = false;
    int pad_t = 0, pad_l = 0, pad_b = 0, pad_r = 0;
};

struct TflModel {
    std::string description;
    std::vector<TflTensor> tensors;
    std::vector<TflOp> ops;
    std::vector<int> inputs, outputs;
    // storage for constants that do not live in the caller's blob (ONNX layout conversions, folded weights, widened
    // float16): shared so that copies of the model stay valid
    std::vector<std::shared_ptr<std::vector<uint8_t>>> owned;
    // adds a float32 constant tensor owning a copy of `v`; returns its index
    int add_const_f32(const std::string& name, const std::vector<int>& shape, const std::vector<float>& v);
    int add_const_i32(const std::string& name, const std::vector<int>& shape, const std::vector<int32_t>& v);
};

// Parses `blob`; returns false and fills `err` on malformed input. Never reads out of bounds.
bool parse_tflite(const void* blob, size_t n, TflModel* out, std::string* err);

// Per-operator operand validation (operand counts, required operands present, constant operand dtypes and the weight
// dimensions the planner indexes by): a well-framed but malformed graph is reported as a model error instead of being
// read out of bounds by the planner.  Runs on the output of parse_tflite / parse_onnx.
bool validate_graph(const TflModel& m, std::string* err);

// Graph rewrites applied before planning (graph_passes.cpp): float16 constants behind DEQUANTIZE widened to float32;
// per-channel constant MUL / ADD / SUB after a convolution or dense layer folded into its weights and bias (unfolded
// batch norm); explicit zero PAD in front of a VALID convolution / pool folded into the op's padding.
bool run_graph_passes(TflModel* m, std::string* err);

}  // namespace bnhip
