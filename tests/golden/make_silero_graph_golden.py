"""Writes tests/golden/silero_graph_nodes.json: the TOPOLOGY of the reference's VAD asset
(/root/reference/faster_whisper/assets/silero_vad_v6.onnx) — its 25 nodes (operator, inputs, outputs, attributes) and
its small integer / scalar constants (pad widths, slice bounds, axes, the exponent), NOT its weights — so that the
generic graph executor (oracle/onnx_exec.py) can be run against the hand restatement (oracle/silero.py) with random
weights on a box that has no copy of the asset.  Build container only:

    python tests/golden/make_silero_graph_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from faster_whisper_amd import onnx_lite  # noqa: E402

ASSET = "/root/reference/faster_whisper/assets/silero_vad_v6.onnx"


def main():
    nodes, inits, ins, outs = onnx_lite.load(ASSET)
    consts = {k: {"dtype": str(v.dtype), "shape": list(v.shape), "values": np.asarray(v).reshape(-1).tolist()}
              for k, v in inits.items() if v.size <= 8 and not k.startswith("decoder.")}
    weights = {k: list(v.shape) for k, v in inits.items() if k not in consts}
    doc = {"source": "faster_whisper/assets/silero_vad_v6.onnx (node list and small constants; no weights)",
           "inputs": ins, "outputs": outs, "nodes": nodes, "constants": consts, "weight_shapes": weights}
    with open(os.path.join(HERE, "silero_graph_nodes.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(len(nodes), "nodes,", len(consts), "constants,", len(weights), "weight tensors")


if __name__ == "__main__":
    main()
