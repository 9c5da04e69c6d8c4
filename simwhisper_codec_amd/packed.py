"""Pre-packed checkpoints.

`AudioCodec._pack` turns the reference's 711-tensor state_dict (model.py:375-396) into GEMM-ready device operands: weight
norm folded (g v / |v|), conv kernels re-laid as [Cout][tap][Cin], frame-stack columns re-ordered, per-tensor power-of-two
scales chosen (one host sync per split-f16 / fp8 tensor), casts to bf16 / split-f16 / e4m3, the fused ConvNeXt operand
streams, the DFT / mel / inverse-DFT tables.  A packed file stores the RESULT of that pass for one precision preset:

    tools/pack_checkpoint.py --fold --precision mixed ...   ->   one .safetensors file (no pickle; memory-mappable)
    AudioCodec.load_from_checkpoint(config, packed_file)    ->   tensors go file -> device, `_pack` never runs

The structure (nested lists / dicts / tuples / small objects, scalars, dtypes) travels as a JSON skeleton in the file's
metadata; every tensor is a flat safetensors entry.  Loading executes nothing from the file.
"""
import json

import torch

FORMAT = "simwhisper-codec packed operands v1"  # optional parts ("extra") are additional skeletons in the metadata
_DT = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16,
       "float8_e4m3fn": torch.float8_e4m3fn, "int32": torch.int32, "int64": torch.int64, "uint8": torch.uint8}


def flatten(obj, classes):
    """obj -> (skeleton (JSON-able), {name: tensor}).  `classes`: {name: class} of the small container classes that may
    appear (their __dict__ / __slots__ are walked)."""
    tensors = {}
    by_cls = {c: n for n, c in classes.items()}

    def walk(o):
        if torch.is_tensor(o):
            name = f"t{len(tensors)}"
            t = o.detach()
            if t.dtype == torch.float8_e4m3fn:  # stored as bytes (the safetensors build here may predate fp8 dtypes)
                tensors[name] = t.contiguous().view(torch.uint8).cpu()
                return {"__t__": name, "view": "float8_e4m3fn"}
            tensors[name] = t.contiguous().cpu()
            return {"__t__": name}
        if isinstance(o, torch.dtype):
            return {"__dtype__": str(o).replace("torch.", "")}
        if isinstance(o, (bool, int, float, str)) or o is None:
            return o
        if isinstance(o, tuple):
            return {"__tuple__": [walk(v) for v in o]}
        if isinstance(o, list):
            return [walk(v) for v in o]
        if isinstance(o, dict):
            return {"__dict__": {str(k): walk(v) for k, v in o.items()}}
        if type(o) in by_cls:
            names = getattr(type(o), "__slots__", None) or sorted(vars(o))
            return {"__obj__": by_cls[type(o)], "fields": {n: walk(getattr(o, n)) for n in names if hasattr(o, n)}}
        raise TypeError(f"cannot pack {type(o)}")

    return walk(obj), tensors


def unflatten(skel, get_tensor, classes):
    def build(o):
        if isinstance(o, list):
            return [build(v) for v in o]
        if not isinstance(o, dict):
            return o
        if "__t__" in o:
            t = get_tensor(o["__t__"])
            return t.view(_DT[o["view"]]) if "view" in o else t
        if "__dtype__" in o:
            return _DT[o["__dtype__"]]
        if "__tuple__" in o:
            return tuple(build(v) for v in o["__tuple__"])
        if "__dict__" in o:
            return {k: build(v) for k, v in o["__dict__"].items()}
        if "__obj__" in o:
            cls = classes[o["__obj__"]]
            inst = cls.__new__(cls)
            for n, v in o["fields"].items():
                setattr(inst, n, build(v))
            return inst
        raise TypeError(f"bad skeleton node {list(o)[:3]}")

    return build(skel)


def save(path, packed, classes, meta, extra=None):
    """extra: {part name: object} stored beside the main operands under their own skeletons (tensor names prefixed with the
    part name); load() never touches them, load_extra() reads one part."""
    from safetensors.torch import save_file
    skel, tensors = flatten(packed, classes)
    md = {"format": FORMAT, "skeleton": json.dumps(skel), "meta": json.dumps(meta)}
    for part, obj in (extra or {}).items():
        sk, ts = flatten(obj, classes)
        ren = {old: f"{part}.{old}" for old in ts}
        md[f"skeleton.{part}"] = json.dumps(_rename(sk, ren))
        tensors.update({ren[k]: v for k, v in ts.items()})
    save_file(tensors, path, metadata=md)
    return len(tensors), sum(t.numel() * t.element_size() for t in tensors.values())


def _rename(skel, ren):
    if isinstance(skel, list):
        return [_rename(v, ren) for v in skel]
    if isinstance(skel, dict):
        if "__t__" in skel:
            return dict(skel, __t__=ren[skel["__t__"]])
        return {k: _rename(v, ren) for k, v in skel.items()}
    return skel


def peek(path):
    """metadata of a safetensors file if it is a packed-operand file, else None (reads the header only)."""
    from safetensors import safe_open
    with safe_open(path, framework="pt", device="cpu") as f:
        md = f.metadata() or {}
    if md.get("format") != FORMAT:
        return None
    return json.loads(md["meta"])


def load_extra(path, device, classes, part):
    """the object stored under `part` by save(..., extra=...), tensors on `device`; None if the file has no such part"""
    from safetensors import safe_open
    with safe_open(path, framework="pt", device=str(device)) as f:
        md = f.metadata() or {}
        if md.get("format") != FORMAT or f"skeleton.{part}" not in md:
            return None
        return unflatten(json.loads(md[f"skeleton.{part}"]), f.get_tensor, classes)


def load(path, device, classes):
    from safetensors import safe_open
    with safe_open(path, framework="pt", device=str(device)) as f:
        md = f.metadata() or {}
        if md.get("format") != FORMAT:
            raise ValueError(f"{path} is not a packed-operand file")
        packed = unflatten(json.loads(md["skeleton"]), f.get_tensor, classes)
    return packed, json.loads(md["meta"])
