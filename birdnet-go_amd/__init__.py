"""birdnet-go_amd: MI355X-native BirdNET inference hot path behind birdnet-go's
`inference.Classifier` seam (reference `internal/inference/backend.go:8-29`).

Only what the hot path needs lives here: the HIP/C++ engine + C ABI (`csrc/`, built into
`lib/libbnhip.so`), the host-side mirror of the reference interface (`host.py`), the model
container tooling (`tflite_schema.py`, `tflite_build.py`, `synth_model.py`) and WAV ingest.
The directory name carries a hyphen (task contract); import it as `birdnet_go_amd` via the
alias module at the repo root.
"""
from . import tflite_schema, flatbuf_writer, tflite_build, synth_model, build, host, shard, wav, results  # noqa: F401

__all__ = ["tflite_schema", "flatbuf_writer", "tflite_build", "synth_model", "build", "host", "shard", "wav", "results"]
