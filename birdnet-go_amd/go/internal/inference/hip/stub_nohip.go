//go:build !hip

// Stub for builds without the `hip` tag (mirrors internal/inference/openvino/stub_noopenvino.go):
// every constructor reports ErrHIPUnavailable so callers keep their existing backend; the types exist so that code
// referring to them compiles either way.
package hip

import "errors"

const Supported = false

var ErrHIPUnavailable = errors.New("hip: backend unavailable")

type Options struct {
	Devices                       []int
	MaxBatch                      int
	Precision                     string
	LogitsOutput, EmbeddingOutput int
	StrictF32                     bool
}

type Classifier struct{}

func Init(string) error                                               { return ErrHIPUnavailable }
func NewClassifier([]byte, ...int) (*Classifier, error)               { return nil, ErrHIPUnavailable }
func NewClassifierWithOptions([]byte, Options) (*Classifier, error)   { return nil, ErrHIPUnavailable }
func (*Classifier) Predict([]float32) ([]float32, error)              { return nil, ErrHIPUnavailable }
func (*Classifier) PredictBatch([]float32, int) ([]float32, error)    { return nil, ErrHIPUnavailable }
func (*Classifier) PredictPCM16([]byte, int) ([]float32, error)       { return nil, ErrHIPUnavailable }
func (*Classifier) NumSpecies() int                                   { return 0 }
func (*Classifier) Close()                                            {}
func (*Classifier) PredictWithEmbeddings([]float32) ([]float32, []float32, error) {
	return nil, nil, ErrHIPUnavailable
}
func (*Classifier) PredictTopK([]float32, int, int, float64) ([]float32, []int32, error) {
	return nil, nil, ErrHIPUnavailable
}
func (*Classifier) PredictTopKSoftmax([]float32, int, int) ([]float32, []int32, error) {
	return nil, nil, ErrHIPUnavailable
}

type WindowAssembler struct{}

func NewWindowAssembler(int, int, int) (*WindowAssembler, error)             { return nil, ErrHIPUnavailable }
func (*WindowAssembler) AddSource(string, int) (int, error)                  { return -1, ErrHIPUnavailable }
func (*WindowAssembler) RemoveSource(int) error                              { return nil }
func (*WindowAssembler) Write(int, []byte) error                             { return ErrHIPUnavailable }
func (*WindowAssembler) Collect() ([]int, []byte, error)                     { return nil, nil, ErrHIPUnavailable }
func (*WindowAssembler) OverwriteStats(int) (uint64, uint64, error)          { return 0, 0, ErrHIPUnavailable }
func (*WindowAssembler) Reset(int) error                                     { return nil }
func (*WindowAssembler) WindowBytes() int                                    { return 0 }
func (*WindowAssembler) Pinned() bool                                        { return false }
func (*WindowAssembler) Close()                                              {}
func (*Classifier) PredictWindows(*WindowAssembler) ([]int, []byte, []float32, error) {
	return nil, nil, nil, ErrHIPUnavailable
}

func (*Classifier) PredictWindowsTopK(*WindowAssembler, int, float64) ([]int, []byte, []float32, []int32, error) {
	return nil, nil, nil, nil, ErrHIPUnavailable
}

type CustomClassifier struct{}

func NewCustomClassifier([]byte, []string, ...int) (*CustomClassifier, error) { return nil, ErrHIPUnavailable }
func (*CustomClassifier) PredictEmbedding([]float32) ([]float32, error)      { return nil, ErrHIPUnavailable }
func (*CustomClassifier) NumClasses() int                                    { return 0 }
func (*CustomClassifier) InputDim() int                                      { return 0 }
func (*CustomClassifier) Labels() []string                                   { return nil }
func (*CustomClassifier) Close()                                             {}

type RangeFilter struct{}

func NewRangeFilter([]byte, ...int) (*RangeFilter, error)                { return nil, ErrHIPUnavailable }
func (*RangeFilter) Predict(float32, float32, float32) ([]float32, error) { return nil, ErrHIPUnavailable }
func (*RangeFilter) PredictBatch([]float32, int) ([]float32, error)      { return nil, ErrHIPUnavailable }
func (*RangeFilter) NumSpecies() int                                     { return 0 }
func (*RangeFilter) Close()                                              {}

type USFilterConfig struct{ FFTSize, HopSize, FrequencySplitHz int }

func ComputeUSFrameCV([]float64, int, USFilterConfig, int) (float64, bool, error) {
	return 0, false, ErrHIPUnavailable
}

type Resampler struct{}

func NewResampler(int, int, int) (*Resampler, error)         { return nil, ErrHIPUnavailable }
func ResampleBytes([]byte, int, int, int) ([]byte, error)    { return nil, ErrHIPUnavailable }
func (*Resampler) EstimateOutputBytes(int) int               { return 0 }
func (*Resampler) ResampleTo([]byte, []byte) (int, error)    { return 0, ErrHIPUnavailable }
func (*Resampler) ResampleInto([]byte) ([]byte, error)       { return nil, ErrHIPUnavailable }
func (*Resampler) Flush() ([]byte, error)                    { return nil, ErrHIPUnavailable }
func (*Resampler) FromRate() int                             { return 0 }
func (*Resampler) ToRate() int                               { return 0 }
func (*Resampler) Close() error                              { return nil }
func (*Resampler) String() string                            { return "Resampler(hip: unavailable)" }
