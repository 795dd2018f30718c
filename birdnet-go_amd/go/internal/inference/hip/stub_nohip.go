//go:build !hip

// Stub for builds without the `hip` tag (mirrors internal/inference/openvino/stub_noopenvino.go):
// every constructor reports ErrHIPUnavailable so callers keep their existing backend.
package hip

import "errors"

const Supported = false

var ErrHIPUnavailable = errors.New("hip: backend unavailable")

type Classifier struct{}

func Init(string) error                                   { return ErrHIPUnavailable }
func NewClassifier([]byte, int) (*Classifier, error)      { return nil, ErrHIPUnavailable }
func (*Classifier) Predict([]float32) ([]float32, error)  { return nil, ErrHIPUnavailable }
func (*Classifier) PredictWithEmbeddings([]float32) ([]float32, []float32, error) {
	return nil, nil, ErrHIPUnavailable
}
func (*Classifier) PredictBatch([]float32, int) ([]float32, error) { return nil, ErrHIPUnavailable }
func (*Classifier) NumSpecies() int                                { return 0 }
func (*Classifier) Close()                                         {}
