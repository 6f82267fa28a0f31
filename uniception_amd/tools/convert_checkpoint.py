"""Original CroCo v2 / DUSt3R / MASt3R checkpoints -> the state_dict layout of ``uniception_amd.models.factory.DUSt3R``
(and of its sub-modules), so real released weights can be put through the HIP path.

Restates the key maps of the reference's converter (examples/models/dust3r/convert_dust3r_weights_to_uniception.py:20-213):
    decoder_embed.*                      -> info_sharing.proj_embed.*                          (:38-40)
    dec_blocks.N.*  / dec_blocks2.N.*    -> info_sharing.multi_view_branches.{0,1}.N.*         (:41-46); a checkpoint without
                                            dec_blocks2 (CroCo) gets view 0's blocks duplicated (:28-34)
    dec_norm.*                           -> info_sharing.norm.*                                (:47-49)
    downstream_head{h}.dpt.<feature>     -> dpt_feature_head{h}.<feature>                      (:76-79, :176-183)
    downstream_head{h}.dpt.head.{0,2,4}  -> dpt_regressor_head{h}.{conv1, conv2.0, conv2.2}    (:95-109)
    downstream_head{h}.proj.{weight,bias}-> head{h}.linear.{weight [N,K,1,1], bias}            (:134-151: Linear -> 1x1 Conv2d)
    MASt3R's downstream_head{h}.head_local_features.* has no counterpart and is dropped         (:176-178, :198-202)
and, beyond that script (it only extracts decoder and heads):
    patch_embed.* / enc_blocks.* / enc_norm.*  -> encoder.*  (the CroCoEncoder checkpoint format of encoders/croco.py:101-111)
The reference's script targets an older DPTFeature (it loads the original `act_postprocess.*` names with strict=True, which
today's module — `input_process.i.0.j`, aliased `scratch.layer_rn.i` / `input_process.i.1` — rejects); the maps below produce
today's names INCLUDING every alias key, so ``DUSt3R.load_state_dict(converted, strict=True)`` succeeds:
    dpt.act_postprocess.i.j.X            -> input_process.i.0.j.X
    dpt.scratch.layer{i}_rn.weight       -> scratch.layer{i}_rn.weight == scratch.layer_rn.{i-1}.weight == input_process.{i-1}.1.weight
    dpt.scratch.refinenet4.resConfUnit1  -> dropped (the module deletes that unused unit for DDP, prediction_heads/dpt.py:82-83)
    dpt_feature_head{h}.* == head{h}.0.*,  dpt_regressor_head{h}.* == head{h}.1.*              (factory/dust3r.py:166-192)
Real checkpoints cannot be fetched in the build environment; the maps are exercised on synthetic original-format
checkpoints (tests/test_convert_checkpoint.py: round trip through `uniception_to_original`, strict load, equal outputs).

    python -m uniception_amd.tools.convert_checkpoint ORIGINAL.pth OUT.pth [--per-module-dir DIR]
"""
import argparse
import os
import re
from typing import Dict, Optional, Tuple

import torch

REG_HEAD_MAP = {"0": "conv1", "2": "conv2.0", "4": "conv2.2"}
_DROPPED_PREFIXES = ("mask_token", "dec_pos_embed", "enc_pos_embed", "prediction_head.")


def detect_head_type(sd: Dict[str, torch.Tensor]) -> Optional[str]:
    if any(k.startswith("downstream_head1.dpt.") for k in sd):
        return "dpt"
    if any(k.startswith("downstream_head1.proj.") for k in sd):
        return "linear"
    return None


def _dpt_feature_keys(rest: str):
    """Names (relative to a DPTFeature) that an original `dpt.<rest>` entry fills — aliases included — or () when dropped."""
    m = re.match(r"act_postprocess\.(\d+)\.(\d+)\.(.+)$", rest)
    if m:
        return (f"input_process.{m.group(1)}.0.{m.group(2)}.{m.group(3)}",)
    m = re.match(r"scratch\.layer(\d)_rn\.(.+)$", rest)
    if m:
        i = int(m.group(1))
        return (f"scratch.layer{i}_rn.{m.group(2)}", f"scratch.layer_rn.{i - 1}.{m.group(2)}", f"input_process.{i - 1}.1.{m.group(2)}")
    if rest.startswith("scratch.refinenet4.resConfUnit1."):
        return ()
    if rest.startswith("scratch."):
        return (rest,)
    raise KeyError(f"unexpected DPT key dpt.{rest}")


def original_to_uniception(sd: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], Dict[str, str]]:
    """(state_dict in the DUSt3R factory's namespace incl. alias keys, {original key: reason} for everything dropped)."""
    out, dropped = {}, {}
    two_decoders = any(k.startswith("dec_blocks2.") for k in sd)
    for k, v in sd.items():
        if k.startswith(("patch_embed.", "enc_blocks.", "enc_norm.")):
            out["encoder." + k] = v
        elif k.startswith("decoder_embed."):
            out["info_sharing.proj_embed." + k[len("decoder_embed."):]] = v
        elif k.startswith("dec_blocks2."):
            out["info_sharing.multi_view_branches.1." + k[len("dec_blocks2."):]] = v
        elif k.startswith("dec_blocks."):
            out["info_sharing.multi_view_branches.0." + k[len("dec_blocks."):]] = v
            if not two_decoders:   # CroCo: one decoder, both views start from it
                out["info_sharing.multi_view_branches.1." + k[len("dec_blocks."):]] = v
        elif k.startswith("dec_norm."):
            out["info_sharing.norm." + k[len("dec_norm."):]] = v
        elif re.match(r"downstream_head[12]\.", k):
            h = k[len("downstream_head")]
            rest = k[len("downstream_headN."):]
            if rest.startswith("head_local_features."):
                dropped[k] = "MASt3R local-feature head: no counterpart in the DUSt3R factory"
            elif rest.startswith("proj."):
                t = v.reshape(v.shape[0], v.shape[1], 1, 1) if rest == "proj.weight" else v
                out[f"head{h}.linear.{rest[len('proj.'):]}"] = t
            elif rest.startswith("dpt.head."):
                idx, leaf = rest[len("dpt.head."):].split(".", 1)
                if idx not in REG_HEAD_MAP:
                    raise KeyError(f"unexpected regressor layer in {k}")
                for pre in (f"dpt_regressor_head{h}.", f"head{h}.1."):
                    out[pre + f"{REG_HEAD_MAP[idx]}.{leaf}"] = v
            elif rest.startswith("dpt."):
                names = _dpt_feature_keys(rest[len("dpt."):])
                if not names:
                    dropped[k] = "refinenet4.resConfUnit1 is unused and deleted by the module"
                for n in names:
                    out[f"dpt_feature_head{h}.{n}"] = v
                    out[f"head{h}.0.{n}"] = v
            else:
                raise KeyError(f"unexpected head key {k}")
        elif k.startswith(_DROPPED_PREFIXES):
            dropped[k] = "not part of the two-view regression path"
        else:
            raise KeyError(f"unexpected key {k} in an original CroCo / DUSt3R / MASt3R checkpoint")
    return out, dropped


def uniception_to_original(sd: Dict[str, torch.Tensor], two_decoders: bool = True) -> Dict[str, torch.Tensor]:
    """Inverse map (canonical names only): a DUSt3R-factory state_dict written with the original checkpoint's key names.
    Used to synthesize original-format checkpoints for tests and to export weights trained here."""
    out = {}
    for k, v in sd.items():
        if k.startswith("encoder."):
            out[k[len("encoder."):]] = v
        elif k.startswith("info_sharing.proj_embed."):
            out["decoder_embed." + k[len("info_sharing.proj_embed."):]] = v
        elif k.startswith("info_sharing.multi_view_branches.0."):
            out["dec_blocks." + k[len("info_sharing.multi_view_branches.0."):]] = v
        elif k.startswith("info_sharing.multi_view_branches.1."):
            if two_decoders:
                out["dec_blocks2." + k[len("info_sharing.multi_view_branches.1."):]] = v
        elif k.startswith("info_sharing.norm."):
            out["dec_norm." + k[len("info_sharing.norm."):]] = v
        elif re.match(r"dpt_feature_head[12]\.", k):
            h, rest = k[len("dpt_feature_head")], k[len("dpt_feature_headN."):]
            m = re.match(r"input_process\.(\d+)\.0\.(\d+)\.(.+)$", rest)
            if m:
                out[f"downstream_head{h}.dpt.act_postprocess.{m.group(1)}.{m.group(2)}.{m.group(3)}"] = v
            elif re.match(r"scratch\.layer_rn\.|input_process\.\d+\.1\.", rest):
                continue   # aliases of scratch.layer{i}_rn
            else:
                out[f"downstream_head{h}.dpt.{rest}"] = v
        elif re.match(r"dpt_regressor_head[12]\.", k):
            h, rest = k[len("dpt_regressor_head")], k[len("dpt_regressor_headN."):]
            inv = {v_: k_ for k_, v_ in REG_HEAD_MAP.items()}
            name, leaf = rest.rsplit(".", 1)
            out[f"downstream_head{h}.dpt.head.{inv[name]}.{leaf}"] = v
        elif re.match(r"head[12]\.linear\.", k):
            h, leaf = k[len("head")], k.rsplit(".", 1)[1]
            out[f"downstream_head{h}.proj.{leaf}"] = v.reshape(v.shape[0], v.shape[1]) if leaf == "weight" else v
        elif re.match(r"head[12]\.[01]\.", k):
            continue       # aliases of dpt_feature_head / dpt_regressor_head
        else:
            raise KeyError(f"unexpected key {k} in a DUSt3R factory state_dict")
    return out


def split_modules(converted: Dict[str, torch.Tensor], data_norm_type: str = "dust3r", patch_embed_cls: str = "PatchEmbedDust3R"):
    """Per-module checkpoints in the formats the modules' `pretrained_checkpoint_path` arguments read
    (encoders/croco.py:101-111: {"model", "data_norm_type", "patch_embed_cls"}; the others {"model"})."""
    def sub(prefix):
        return {k[len(prefix):]: v for k, v in converted.items() if k.startswith(prefix)}
    mods = {"encoder": {"model": sub("encoder."), "data_norm_type": data_norm_type, "patch_embed_cls": patch_embed_cls},
            "info_sharing": {"model": sub("info_sharing.")}}
    for h in ("1", "2"):
        if any(k.startswith(f"dpt_feature_head{h}.") for k in converted):
            mods[f"dpt_feature_head{h}"] = {"model": sub(f"dpt_feature_head{h}.")}
            mods[f"dpt_regressor_head{h}"] = {"model": sub(f"dpt_regressor_head{h}.")}
        elif any(k.startswith(f"head{h}.linear.") for k in converted):
            mods[f"linear_feature_head{h}"] = {"model": sub(f"head{h}.")}
    return {k: v for k, v in mods.items() if v["model"]}


def load_original_checkpoint(model: torch.nn.Module, original_sd: Dict[str, torch.Tensor], strict: bool = True):
    """Convert an original-format state_dict and load it into a ``DUSt3R`` factory model (strict by default)."""
    converted, _ = original_to_uniception(original_sd)
    res = model.load_state_dict(converted, strict=strict)
    from .. import engine
    engine.bump_weight_epoch()
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("original")
    ap.add_argument("out")
    ap.add_argument("--per-module-dir", default=None, help="also write encoder / info_sharing / head checkpoints there")
    a = ap.parse_args(argv)
    ckpt = torch.load(a.original, map_location="cpu", weights_only=False)
    sd = ckpt["model"] if "model" in ckpt else ckpt
    converted, dropped = original_to_uniception(sd)
    for k, why in dropped.items():
        print(f"dropped {k}: {why}")
    torch.save({"model": converted, "head_type": detect_head_type(sd)}, a.out)
    print(f"wrote {a.out}: {len(converted)} entries ({len(sd)} in the original)")
    if a.per_module_dir:
        os.makedirs(a.per_module_dir, exist_ok=True)
        for name, c in split_modules(converted).items():
            torch.save(c, os.path.join(a.per_module_dir, name + ".pth"))
            print(f"wrote {os.path.join(a.per_module_dir, name + '.pth')}")


if __name__ == "__main__":
    main()
