"""Host-side on-disk formats of the reference's data pipeline: `.qnt.pt`, `.phn.txt`, symmap.
Each check cites the reference statement whose behaviour it pins."""
import json
import sys
from pathlib import Path

import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tts-with-diffusion-model_amd"))
from vall_e import formats  # noqa: E402


def test_quants_round_trip(tmp_path):
    # emb/qnt.py:93 saves encode()'s [1, q, t]; data.py:31-37 loads `[0].t()` -> (t, q)
    codes = torch.randint(0, 1024, (1, 8, 37), dtype=torch.int64)
    torch.save(codes, tmp_path / "a.qnt.pt")
    tq = formats.load_quants(tmp_path / "a.qnt.pt")
    assert tq.shape == (37, 8) and tq.dtype == torch.int64 and torch.equal(tq, codes[0].t())
    formats.save_quants(tq, tmp_path / "b.qnt.pt")
    assert torch.equal(torch.load(tmp_path / "b.qnt.pt"), codes)
    formats.save_quants(tq[:, 0], tmp_path / "c.qnt.pt")          # level 0 only, as the D3PM stage alone produces
    assert torch.load(tmp_path / "c.qnt.pt").shape == (1, 1, 37)
    torch.save(codes[0], tmp_path / "bad.qnt.pt")
    with pytest.raises(ValueError):
        formats.load_quants(tmp_path / "bad.qnt.pt")


def test_phones_and_symmap(tmp_path):
    (tmp_path / "u.phn.txt").write_text("HH AH0 L OW1 _ W ER1 L D\n", encoding="utf8")
    phones = formats.read_phones(tmp_path / "u.phn.txt")
    assert phones[0] == "<s>" and phones[-1] == "</s>" and phones[1:-1] == "HH AH0 L OW1 _ W ER1 L D".split()   # data.py:40-45
    symmap = formats.build_symmap([phones])
    assert min(symmap.values()) == 1 and sorted(symmap.values()) == list(range(1, len(symmap) + 1))              # data.py:125-127
    assert list(symmap) == sorted(symmap)
    ids = formats.phones_to_ids(phones, symmap)
    assert ids.dtype == torch.int64 and ids.min() >= 1 and len(ids) == len(phones)
    (tmp_path / "symmap.json").write_text(json.dumps(symmap), encoding="utf8")
    assert formats.load_symmap(tmp_path / "symmap.json") == symmap
    with pytest.raises(KeyError):
        formats.phones_to_ids(["<s>", "ZZ9"], symmap)
    (tmp_path / "zero.json").write_text(json.dumps({"a": 0}), encoding="utf8")
    with pytest.raises(ValueError):
        formats.load_symmap(tmp_path / "zero.json")                # id 0 is padding (ar_discrete.py:210 padding_idx=0)


def test_cli_rejects_ambiguous_text(tmp_path, capsys):
    from vall_e import __main__ as cli
    torch.save(torch.zeros(1, 8, 4, dtype=torch.int64), tmp_path / "p.qnt.pt")
    with pytest.raises(SystemExit):
        cli.main([str(tmp_path / "o.qnt.pt"), "--prompt-qnt", str(tmp_path / "p.qnt.pt")])
    with pytest.raises(SystemExit):
        cli.main([str(tmp_path / "o.qnt.pt"), "--prompt-qnt", str(tmp_path / "p.qnt.pt"), "--phn-file", "x.phn.txt"])


def test_reads_the_references_own_files_like_the_references_own_loaders():
    """tests/golden/formats/ holds `.qnt.pt` / `.phn.txt` files written the way the reference's front-ends write them and
    `expected.json` = what the reference's OWN `_load_quants` / `_get_phones` / VALLEDatset symmaps (data.py:31-45,119-134)
    made of them (tests/golden/make_golden.py:gen_formats, run with the imported reference).  formats.py must agree."""
    root = Path(__file__).resolve().parent / "golden" / "formats"
    expect = json.loads((root / "expected.json").read_text())
    phone_lists = []
    for rel, want in expect["utterances"].items():
        q = formats.load_quants(root / (rel + ".qnt.pt"))
        assert q.dtype == torch.int64 and q.tolist() == want["quants_t_q"]
        phones = formats.read_phones(root / (rel + ".phn.txt"))
        assert phones == want["phones"]
        phone_lists.append(phones)
    symmap = formats.build_symmap(phone_lists)
    assert symmap == expect["phone_symmap"]                       # ids from 1, sorted symbols, <s> and </s> included
    for rel, want in expect["utterances"].items():
        ids = formats.phones_to_ids(formats.read_phones(root / (rel + ".phn.txt")), symmap)
        assert ids.tolist() == want["text_ids"]                   # data.py:166 `map(self.phone_symmap.get, _get_phones(path))`


def test_symmaps_ride_in_the_state_dict_only_when_set():
    """export.py:18-19 attaches phone_symmap / spkr_symmap to the module; __main__.py:56 reads ar.phone_symmap.  Here they are
    module attributes that round-trip through state_dict under ONE extra key, absent when unset (so the 271-tensor reference
    layout is untouched and an upstream state dict loads strictly)."""
    from vall_e.vall_e import AR, NAR
    for make in (AR.reference_native, lambda: NAR(1024, d_model=64, n_heads=2, n_layers=1)):
        m = make()
        assert m.phone_symmap == {} and m.spkr_symmap == {}
        plain = m.state_dict()
        assert "_symmaps" not in plain
        m.phone_symmap, m.spkr_symmap = {"<s>": 2, "AH0": 4}, {"spk_a": 0}
        sd = m.state_dict()
        assert set(sd) - set(plain) == {"_symmaps"}
        m2 = make()
        m2.load_state_dict(sd)                                    # strict
        assert m2.phone_symmap == {"<s>": 2, "AH0": 4} and m2.spkr_symmap == {"spk_a": 0}
        m3 = make()
        m3.load_state_dict(plain)                                 # an upstream-style dict: still strict, maps stay empty
        assert m3.phone_symmap == {}


def test_cli_upstream_form_names_the_missing_front_end(tmp_path):
    """`python -m vall_e TEXT REFERENCE OUT` (reference __main__.py:44-51): accepted; its third-party front-ends are
    imported lazily, and where g2p_en / encodec are not installed the error says so instead of falling back."""
    import importlib.util
    from vall_e import __main__ as cli
    with pytest.raises(SystemExit):                                   # TEXT needs a phone symmap from somewhere
        cli.main(["hello world", str(tmp_path / "ref.wav"), str(tmp_path / "out.wav")])
    (tmp_path / "symmap.json").write_text(json.dumps({"HH": 1, "_": 2}))
    if importlib.util.find_spec("g2p_en") is None:
        with pytest.raises(RuntimeError, match="g2p_en"):
            cli.main(["hello world", str(tmp_path / "ref.wav"), str(tmp_path / "out.wav"), "--symmap", str(tmp_path / "symmap.json")])
