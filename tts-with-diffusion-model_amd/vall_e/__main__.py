"""`python -m vall_e` -- caller of the D3PM sampler with the reference CLI's shape
(/root/reference/vall_e/__main__.py:44-73: TEXT REFERENCE OUT [--ar-ckpt] [--nar-ckpt] [--device]).

Two forms:

    python -m vall_e TEXT REFERENCE.wav OUT.wav [--ar-ckpt zoo/ar.pt] [--nar-ckpt zoo/nar.pt] [--device cuda]

is upstream's command line.  Its front-ends are third-party and need downloads (g2p_en for phonemes, EnCodec for the
prompt and for decoding): they are imported lazily (frontends.py) and a missing one is an error that names it.
The second form takes what those front-ends write (formats.py) and writes what EnCodec's decoder reads:

    python -m vall_e --phn-file utt.phn.txt --symmap symmap.json --prompt-qnt prompt.qnt.pt \
                     --ar-ckpt ar_state_dict.pt --nar-ckpt nar_state_dict.pt out.qnt.pt

  --phonemes    space-separated phoneme ids, or
  --phn-file    a `.phn.txt` file of phone symbols, mapped through `ar.phone_symmap` like upstream (__main__.py:56,61) when
                the checkpoint carries it (state dict key `_symmaps`, or a tools/convert_upstream_pickle.py export), else
                through --symmap (a JSON {symbol: id})
  --prompt-qnt  a `.qnt.pt` file as written by the reference's emb/qnt.py:68,93 (int64 [1, 8, t])
  --nar-ckpt    state_dict of the stock NAR model: levels 1..7 are filled in and `out` is a `.qnt.pt` [1, 8, t];
                without it `out` holds the level-0 codes only ([1, 1, t])
"""
import argparse
from pathlib import Path

import torch


def main(argv=None):
    ap = argparse.ArgumentParser("D3PM codec-token sampler (MI355X)")
    ap.add_argument("paths", nargs="+", metavar="[TEXT REFERENCE] OUT",
                    help="OUT.qnt.pt (pre-tokenised form), or upstream's TEXT REFERENCE.wav OUT.wav")
    ap.add_argument("--phonemes", default=None)
    ap.add_argument("--phn-file", type=Path, default=None)
    ap.add_argument("--symmap", type=Path, default=None)
    ap.add_argument("--prompt-qnt", type=Path, default=None)
    ap.add_argument("--ar-ckpt", type=Path, default=None, help="state_dict in the reference key layout (random init if absent)")
    ap.add_argument("--nar-ckpt", type=Path, default=None, help="state_dict of the NAR model (vall_e/vall_e/nar.py)")
    ap.add_argument("--nar-model", default="nar", help="registry name of the NAR model: nar, nar-half, nar-quarter")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16", "float32"])
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--native", action="store_true", help="the shape upstream's class really builds (d=32, 16 heads, 8 blocks)")
    args = ap.parse_args(argv)

    from . import formats
    from .vall_e import AR, get_model
    if len(args.paths) not in (1, 3):
        ap.error("give OUT, or TEXT REFERENCE OUT")
    upstream_form = len(args.paths) == 3
    args.out_path = Path(args.paths[-1])
    if upstream_form:
        if args.phonemes is not None or args.phn_file is not None or args.prompt_qnt is not None:
            ap.error("TEXT REFERENCE OUT takes its phonemes and prompt from TEXT and REFERENCE")
        if args.ar_ckpt is None and args.symmap is None:
            ap.error("TEXT needs a phone symmap: an --ar-ckpt that carries ar.phone_symmap, or --symmap")
        from . import frontends
        phones = frontends.g2p_encode(args.paths[0])                                # emb/g2p.py:24-28
        proms = frontends.encodec_encode_file(args.paths[1], args.device)[0].t().contiguous().long().cpu()   # "1 l t -> t l"
    else:
        if (args.phonemes is None) == (args.phn_file is None):
            ap.error("give exactly one of --phonemes / --phn-file")
        if args.phn_file is not None and args.symmap is None and args.ar_ckpt is None:
            ap.error("--phn-file needs a phone symmap: --symmap, or an --ar-ckpt that carries ar.phone_symmap")
        if args.prompt_qnt is None:
            ap.error("--prompt-qnt is required")
        proms = formats.load_quants(args.prompt_qnt)                               # (t, 8) like data.py:31-37

    dtype = getattr(torch, args.dtype)
    model = AR.reference_native().to(args.device) if args.native else get_model("diffusion")
    if args.ar_ckpt is not None:
        blob = torch.load(args.ar_ckpt, map_location="cpu")
        if isinstance(blob, dict) and "state_dict" in blob:                         # tools/convert_upstream_pickle.py export
            model.load_state_dict(blob["state_dict"])
            model.phone_symmap = dict(blob.get("phone_symmap") or {})
            model.spkr_symmap = dict(blob.get("spkr_symmap") or {})
        else:
            model.load_state_dict(blob)
    model = model.to(dtype).to(args.device)
    if upstream_form or args.phn_file is not None:
        symmap = formats.load_symmap(args.symmap) if args.symmap is not None else model.phone_symmap
        if not symmap:
            ap.error("no phone symmap in the checkpoint: give --symmap")
        phones = phones if upstream_form else formats.read_phones(args.phn_file)
        phns = formats.phones_to_ids(phones, symmap)                                # symmap = ar.phone_symmap, __main__.py:56,61
    else:
        phns = torch.tensor([int(p) for p in args.phonemes.split()], dtype=torch.long)
    codes = model.generate_audio(text_list=[phns], proms_list=[proms], seed=args.seed)
    resps = codes[: model.cfg.n_frames].unsqueeze(-1)                              # __main__.py:64 of the reference
    if args.nar_ckpt is not None:
        nar = get_model(args.nar_model)
        nar.load_state_dict(torch.load(args.nar_ckpt, map_location="cpu"))
        nar = nar.to(dtype).to(args.device)
        resps = nar(text_list=[phns.to(args.device)], proms_list=[proms.to(args.device)], resps_list=[resps],
                    seed=args.seed)[0]
    if upstream_form:
        from . import frontends
        frontends.encodec_decode_to_file(resps, args.out_path, args.device)         # qnt.decode_to_file, __main__.py:72
    else:
        formats.save_quants(resps, args.out_path)
    print(args.out_path, "saved.")


if __name__ == "__main__":
    main()
