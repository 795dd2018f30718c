"""TFLite flatbuffer schema constants (subset) shared by the model writer and readers.

The reference hands the classifier backend an in-memory ``.tflite`` byte slice
(`internal/classifier/birdnet.go:1195-1246` loadModel -> `NewTFLiteClassifier(modelData []byte, ...)`
`internal/inference/tflite/classifier.go:38`).  The container format is therefore part of the
drop-in boundary.  The schema itself belongs to TensorFlow Lite 2.17.1 (third-party, not under
/root/reference; pinned in `Taskfile.yml:6`); the slot numbers below restate its published
`schema.fbs` (field declaration order == vtable slot).
"""

# ---- vtable slots: table name -> {field: slot}
MODEL = dict(version=0, operator_codes=1, subgraphs=2, description=3, buffers=4,
             metadata_buffer=5, metadata=6, signature_defs=7)
SUBGRAPH = dict(tensors=0, inputs=1, outputs=2, operators=3, name=4)
TENSOR = dict(shape=0, type=1, buffer=2, name=3, quantization=4, is_variable=5,
              sparsity=6, shape_signature=7, has_rank=8, variant_tensors=9)
OPERATOR = dict(opcode_index=0, inputs=1, outputs=2, builtin_options_type=3,
                builtin_options=4, custom_options=5, custom_options_format=6,
                mutating_variable_inputs=7, intermediates=8)
OPERATOR_CODE = dict(deprecated_builtin_code=0, custom_code=1, version=2, builtin_code=3)
BUFFER = dict(data=0, offset=1, size=2)

# ---- TensorType
FLOAT32, FLOAT16, INT32, UINT8, INT64, STRING, BOOL, INT16, COMPLEX64, INT8, FLOAT64 = range(11)

# ---- Padding / fused activation
PAD_SAME, PAD_VALID = 0, 1
ACT_NONE, ACT_RELU, ACT_RELU_N1_TO_1, ACT_RELU6, ACT_TANH = 0, 1, 2, 3, 4

# ---- BuiltinOperator codes (subset this engine understands)
OP = dict(
    ADD=0, AVERAGE_POOL_2D=1, CONCATENATION=2, CONV_2D=3, DEPTHWISE_CONV_2D=4, DEQUANTIZE=6, FLOOR=8,
    FULLY_CONNECTED=9, LOGISTIC=14, MAX_POOL_2D=17, MUL=18, RELU=19, RELU_N1_TO_1=20, RELU6=21, RESHAPE=22,
    SOFTMAX=25, TANH=28, PAD=34, GATHER=36, TRANSPOSE=39, MEAN=40, SUB=41, DIV=42, SQUEEZE=43,
    STRIDED_SLICE=45, EXP=47, SPLIT=49, CAST=53, MAXIMUM=55, MINIMUM=57, NEG=59, PADV2=60, SLICE=65, SIN=66,
    EXPAND_DIMS=70, LOG=73, SUM=74, SQRT=75, RSQRT=76, POW=78, REDUCE_PROD=81, REDUCE_MAX=82, REDUCE_MIN=89,
    SQUARE=92, LEAKY_RELU=98, SQUARED_DIFFERENCE=99, ABS=101, CEIL=104, REVERSE_V2=105, COS=108, ELU=111,
    ROUND=116, HARD_SWISH=117, BATCH_MATMUL=126, RFFT2D=131, IMAG=133, REAL=134, COMPLEX_ABS=135, GELU=150,
)
OP_NAME = {v: k for k, v in OP.items()}

# ---- BuiltinOptions union type ids
OPT = dict(
    NONE=0, Conv2DOptions=1, DepthwiseConv2DOptions=2, Pool2DOptions=5, FullyConnectedOptions=8,
    SoftmaxOptions=9, ConcatenationOptions=10, AddOptions=11, ReshapeOptions=17, MulOptions=21,
    PadOptions=22, GatherOptions=23, TransposeOptions=26, ReducerOptions=27, SubOptions=28,
    DivOptions=29, SqueezeOptions=30, StridedSliceOptions=32, CastOptions=37,
    ExpandDimsOptions=52, PowOptions=56, ReverseV2Options=81, HardSwishOptions=91,
    BatchMatMulOptions=101, Rfft2dOptions=105, SplitOptions=35, MaximumMinimumOptions=39, PadV2Options=43,
    SliceOptions=48, LeakyReluOptions=75, SquaredDifferenceOptions=76, GeluOptions=123,
)

# ---- option tables: name -> ordered [(field, kind)]; kind in i8,i32,f32,bool,vec_i32
OPTION_FIELDS = dict(
    Conv2DOptions=[("padding", "i8"), ("stride_w", "i32"), ("stride_h", "i32"),
                   ("fused_activation_function", "i8"), ("dilation_w_factor", "i32"),
                   ("dilation_h_factor", "i32")],
    DepthwiseConv2DOptions=[("padding", "i8"), ("stride_w", "i32"), ("stride_h", "i32"),
                            ("depth_multiplier", "i32"), ("fused_activation_function", "i8"),
                            ("dilation_w_factor", "i32"), ("dilation_h_factor", "i32")],
    Pool2DOptions=[("padding", "i8"), ("stride_w", "i32"), ("stride_h", "i32"),
                   ("filter_width", "i32"), ("filter_height", "i32"),
                   ("fused_activation_function", "i8")],
    FullyConnectedOptions=[("fused_activation_function", "i8"), ("weights_format", "i8"),
                           ("keep_num_dims", "bool"), ("asymmetric_quantize_inputs", "bool")],
    SoftmaxOptions=[("beta", "f32")],
    ConcatenationOptions=[("axis", "i32"), ("fused_activation_function", "i8")],
    AddOptions=[("fused_activation_function", "i8")],
    MulOptions=[("fused_activation_function", "i8")],
    SubOptions=[("fused_activation_function", "i8")],
    DivOptions=[("fused_activation_function", "i8")],
    ReshapeOptions=[("new_shape", "vec_i32")],
    PadOptions=[],
    GatherOptions=[("axis", "i32"), ("batch_dims", "i32")],
    TransposeOptions=[],
    ReducerOptions=[("keep_dims", "bool")],
    SqueezeOptions=[("squeeze_dims", "vec_i32")],
    StridedSliceOptions=[("begin_mask", "i32"), ("end_mask", "i32"), ("ellipsis_mask", "i32"),
                         ("new_axis_mask", "i32"), ("shrink_axis_mask", "i32")],
    CastOptions=[("in_data_type", "i8"), ("out_data_type", "i8")],
    ExpandDimsOptions=[],
    PowOptions=[],
    ReverseV2Options=[],
    HardSwishOptions=[],
    BatchMatMulOptions=[("adj_x", "bool"), ("adj_y", "bool"),
                        ("asymmetric_quantize_inputs", "bool")],
    Rfft2dOptions=[],
    SplitOptions=[("num_splits", "i32")],
    MaximumMinimumOptions=[], PadV2Options=[], SliceOptions=[], SquaredDifferenceOptions=[],
    LeakyReluOptions=[("alpha", "f32")],
    GeluOptions=[("approximate", "bool")],
)

# which options table each builtin op carries
OP_OPTIONS = dict(
    ADD="AddOptions", AVERAGE_POOL_2D="Pool2DOptions", CONCATENATION="ConcatenationOptions",
    CONV_2D="Conv2DOptions", DEPTHWISE_CONV_2D="DepthwiseConv2DOptions", DEQUANTIZE=None,
    FULLY_CONNECTED="FullyConnectedOptions", LOGISTIC=None, MAX_POOL_2D="Pool2DOptions",
    MUL="MulOptions", RELU=None, RELU6=None, RESHAPE="ReshapeOptions", SOFTMAX="SoftmaxOptions",
    PAD="PadOptions", GATHER="GatherOptions", TRANSPOSE="TransposeOptions", MEAN="ReducerOptions",
    SUB="SubOptions", DIV="DivOptions", SQUEEZE="SqueezeOptions",
    STRIDED_SLICE="StridedSliceOptions", CAST="CastOptions", EXPAND_DIMS="ExpandDimsOptions",
    SUM="ReducerOptions", POW="PowOptions", REDUCE_MAX="ReducerOptions",
    REDUCE_MIN="ReducerOptions", REVERSE_V2="ReverseV2Options", HARD_SWISH="HardSwishOptions",
    BATCH_MATMUL="BatchMatMulOptions", RFFT2D="Rfft2dOptions", IMAG=None, REAL=None,
    COMPLEX_ABS=None,
    FLOOR=None, RELU_N1_TO_1=None, TANH=None, EXP=None, SPLIT="SplitOptions", MAXIMUM="MaximumMinimumOptions",
    MINIMUM="MaximumMinimumOptions", NEG=None, PADV2="PadV2Options", SLICE="SliceOptions", SIN=None, LOG=None,
    SQRT=None, RSQRT=None, REDUCE_PROD="ReducerOptions", SQUARE=None, LEAKY_RELU="LeakyReluOptions",
    SQUARED_DIFFERENCE="SquaredDifferenceOptions", ABS=None, CEIL=None, COS=None, ELU=None, ROUND=None,
    GELU="GeluOptions",
)

FILE_IDENTIFIER = b"TFL3"
