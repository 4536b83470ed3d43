"""The place of create_pb.py in this build: turn what the reference's training or export left on disk into ONE weight file
the Detector loads -- without TensorFlow.

    python -m ssd_amd.convert <model.pb | model_dir | model.ckpt-N | export/ | weights.npz> <config.json> <out.npz> [--ema]

create_pb.py:27-85 builds the PREDICT graph, restores the latest checkpoint of `model_dir` and freezes it into
`inference/model.pb`; the graph itself is this library (csrc/), so what remains of the export is the variables: they are read
from the frozen graph (pb_import.py) or straight from the checkpoint / SavedModel variables (ckpt_import.py), checked against
the architecture the config names (variables.variable_shapes), and written as the .npz container of variables.py.
`--ema` takes the moving averages instead of the raw variables (model.py:148-161; create_pb.py itself freezes the raw ones).
`Detector(model_path)` accepts every one of these inputs directly; converting once saves the checkpoint's CRC pass at start-up."""
import sys

from .ckpt_import import load_ckpt_weights, resolve_checkpoint
from .config import load_config
from .pb_import import load_pb_weights
from .variables import load_weights, save_weights, variable_shapes


def load_any(path, params, use_ema=False):
    """{variable name: float32 ndarray} of the architecture `params` names, from any container this build reads."""
    path = str(path)
    if path.endswith(".pb"):
        return load_pb_weights(path, params)
    if path.endswith(".npz"):
        W = load_weights(path)
        shapes = variable_shapes(params)
        for name, shape in shapes.items():
            if name not in W:
                raise KeyError("%s has no variable %r" % (path, name))
            if tuple(W[name].shape) != tuple(shape):
                raise ValueError("variable %r has shape %s, expected %s" % (name, W[name].shape, shape))
        return {name: W[name] for name in shapes}
    if resolve_checkpoint(path) is None:
        raise FileNotFoundError("%s is neither a .pb, a .npz nor a TensorFlow checkpoint / model_dir / SavedModel directory" % path)
    return load_ckpt_weights(path, params, use_ema=use_ema)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ema = "--ema" in argv
    argv = [a for a in argv if a != "--ema"]
    if len(argv) != 3:
        sys.stderr.write(__doc__)
        return 2
    src, config, out = argv
    params = load_config(config)
    W = load_any(src, params, use_ema=ema)
    save_weights(out, W)
    print("%s: %d variables, %.2f M parameters of %s -> %s" % (src, len(W), sum(v.size for v in W.values()) / 1e6, params["backbone"], out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
