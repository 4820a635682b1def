"""Turn an upstream whole-module pickle into a checkpoint this build loads.

The reference ships trained models as `torch.save(model)` of the whole nn.Module with `phone_symmap` / `spkr_symmap`
attached as attributes (/root/reference/vall_e/export.py:14-20); unpickling that needs upstream's classes, so it has to
happen where upstream is importable.  Run this ONCE in the upstream environment:

    python convert_upstream_pickle.py zoo/ar.pt ar_export.pt

and load the result here with `AR.load_exported("ar_export.pt")` / `NAR.load_exported(...)`, or pass it to
`python -m vall_e --ar-ckpt ar_export.pt` (the phone symmap then comes from the checkpoint, as upstream's
__main__.py:56 reads it from the module).  The file holds plain tensors and dicts only:
    {"state_dict": {name: tensor}, "phone_symmap": {symbol: id}, "spkr_symmap": {speaker: id}, "class": "AR"}
This script imports nothing from this repository and nothing from upstream besides what the pickle itself pulls in.
"""
import argparse

import torch


def convert(module) -> dict:
    sd = {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}
    return {"state_dict": sd, "phone_symmap": dict(getattr(module, "phone_symmap", None) or {}),
            "spkr_symmap": dict(getattr(module, "spkr_symmap", None) or {}), "class": type(module).__name__}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("pickle_path")
    ap.add_argument("out_path")
    args = ap.parse_args(argv)
    module = torch.load(args.pickle_path, map_location="cpu", weights_only=False)
    blob = convert(module)
    torch.save(blob, args.out_path)
    print(f"{args.out_path}: {len(blob['state_dict'])} tensors, {len(blob['phone_symmap'])} phones, "
          f"{len(blob['spkr_symmap'])} speakers")


if __name__ == "__main__":
    main()
