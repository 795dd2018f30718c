"""Print the nodes, initialisers and I/O of an ONNX file (through the oracle's independent protobuf reader)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.onnx_interp import OnnxModel


def dump(blob, out=sys.stdout):
    m = OnnxModel(blob)
    print("inputs", m.runtime_inputs, "outputs", m.outputs, file=out)
    for k, a in m.inits.items():
        small = a.reshape(-1)[:6].tolist() if a.size <= 6 else ""
        print(f"  init {k}: {a.dtype} {list(a.shape)} {small}", file=out)
    for op, ins, outs, at in m.nodes:
        ats = {k: (v if not isinstance(v, np.ndarray) else f"tensor{list(v.shape)}{v.reshape(-1)[:4].tolist()}") for k, v in at.items()}
        print(f"  {op}({', '.join(ins)}) -> {', '.join(outs)}  {ats if ats else ''}", file=out)


if __name__ == "__main__":
    dump(open(sys.argv[1], "rb").read())
