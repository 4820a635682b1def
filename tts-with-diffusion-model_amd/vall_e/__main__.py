"""`python -m vall_e` -- caller of the D3PM sampler with the reference CLI's shape
(/root/reference/vall_e/__main__.py:44-73: TEXT REFERENCE OUT [--ar-ckpt] [--device]).

The reference front-ends are third-party and need downloads (g2p_en for phonemes, EnCodec for the prompt and for
decoding) and its second stage is the stock NAR model; none of them is part of this build.  This
entry point therefore takes what those front-ends would produce and writes what the NAR stage would consume:

    python -m vall_e --phonemes "12 7 33 4" --prompt-qnt prompt.qnt.pt --ar-ckpt ar_state_dict.pt out_codes.pt

  --phonemes    space-separated phoneme ids (the reference maps g2p symbols through ar.phone_symmap, 1-based)
  --prompt-qnt  a `.qnt.pt` file as written by the reference's emb/qnt.py:68,93 (int64 [1, 8, t])
  out           torch.save of the level-0 codes, int64 [n_frames] (trimmed) -- `resps_list=[codes.unsqueeze(-1)]`
"""
import argparse
from pathlib import Path

import torch


def main():
    ap = argparse.ArgumentParser("D3PM codec-token sampler (MI355X)")
    ap.add_argument("out_path", type=Path)
    ap.add_argument("--phonemes", required=True)
    ap.add_argument("--prompt-qnt", type=Path, required=True)
    ap.add_argument("--ar-ckpt", type=Path, default=None, help="state_dict in the reference key layout (random init if absent)")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16", "float32"])
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--native", action="store_true", help="the shape upstream's class really builds (d=32, 16 heads, 8 blocks)")
    args = ap.parse_args()

    from .vall_e import AR, get_model
    model = AR.reference_native().to(args.device) if args.native else get_model("diffusion")
    if args.ar_ckpt is not None:
        model.load_state_dict(torch.load(args.ar_ckpt, map_location="cpu"))
    model = model.to(getattr(torch, args.dtype)).to(args.device)
    phns = torch.tensor([int(p) for p in args.phonemes.split()], dtype=torch.long)
    qnt = torch.load(args.prompt_qnt, map_location="cpu")
    proms = qnt[0].t().contiguous().long() if qnt.dim() == 3 else qnt.long()       # (t, 8) like data.py:31-37
    codes = model.generate_audio(text_list=[phns], proms_list=[proms], seed=args.seed)
    torch.save(codes[: model.cfg.n_frames].cpu(), args.out_path)
    print(args.out_path, "saved.")


if __name__ == "__main__":
    main()
