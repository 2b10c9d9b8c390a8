"""Model definitions for the three QuartzNet variants the reference ships, plus a loader
for reference-format YAML files.

The reference reads ``configs/*.yaml`` with ruamel and splats the sections into the module
constructors (infer.py:85-111).  ``load_model_definition`` accepts exactly those files (and the
legacy ``AudioPreprocessing`` section name that configs/quartznet15x5.yaml:16-26 still uses);
``builtin(name)`` rebuilds the same dictionaries from a compact table so that the GPU box --
where /root/reference does not exist -- needs no data files.
"""
import copy

import yaml

# label sets (configs/quartznet12x1_vi.yaml:166-, quartznet15x5.yaml:201-, quartznet12x1.yaml)
LABELS_VI = list(" abcdeghiklmnopqrstuvxyàáâãèéêìíòóôõùúýăđĩũơưạảấầẩẫậắằẳẵặẹẻẽếềểễệỉịọỏốồổỗộớờởỡợụủứừửữựỳỵỷỹ")
LABELS_EN = list(" abcdefghijklmnopqrstuvwxyz'")
LABELS_VI_DIGITS = list(" 0123456789aáàảãạăắằẳẵặâấầẩẫậbcdđeéèẻẽẹfghiíìỉĩịjklmnoóòỏõọôốồổỗộơớờởỡợpqrstuúùủũụưứừửữựvxyýỳỷỹỵz")

_PRE = dict(sample_rate=16000, window_size=0.02, window_stride=0.01, window="hann", normalize="per_feature",
            n_fft=512, features=64, dither=0.00001, pad_to=16, stft_conv=False)


def _blk(filters, repeat, kernel, stride=1, dilation=1, residual=True, separable=True):
    d = dict(filters=filters, repeat=repeat, kernel=[kernel], stride=[stride], dilation=[dilation],
             dropout=0.0, residual=residual)
    if separable:
        d["separable"] = True
    return d


def _quartznet(repeat, with_dilated_tail):
    """B1..B5 x3 QuartzNet body (QuartzNet paper table 1) as the reference YAMLs spell it."""
    blocks = [_blk(256, 1, 33, stride=2, residual=False)]
    for ch, k in ((256, 33), (256, 39), (512, 51), (512, 63), (512, 75)):
        if (ch, k) == (512, 75) and not with_dilated_tail:
            blocks += [_blk(ch, repeat, k)]            # 12x1: a single K=75 block
        else:
            blocks += [_blk(ch, repeat, k) for _ in range(3)]
    if with_dilated_tail:
        blocks.append(_blk(512, 1, 87, dilation=2, residual=False))
    blocks.append(_blk(1024, 1, 1, residual=False, separable=False))
    return blocks


def builtin(name):
    """Model definition dict equal to yaml-loading the reference config of that name -- with one deliberate exception:
    configs/quartznet15x5.yaml:26 says ``stft_conv: true`` (the third-party torch_stft package, absent here and on the
    reference side of the golden generator); the builtin 15x5 keeps ``stft_conv=False`` like the two Vietnamese configs,
    which is how tests/golden/make_golden.py runs the reference.  Loading the real YAML selects the conv-STFT's
    periodic window (frontend_tables.frontend_description)."""
    if name in ("quartznet12x1_vi", "quartznet12x1_vi.yaml"):
        body, labels = _quartznet(1, False), LABELS_VI
    elif name in ("quartznet12x1", "quartznet12x1.yaml"):
        body, labels = _quartznet(1, False), LABELS_VI_DIGITS
    elif name in ("quartznet15x5", "quartznet15x5.yaml"):
        body, labels = _quartznet(5, True), LABELS_EN
    else:
        raise ValueError(f"unknown builtin model {name!r}")
    return {
        "model": name.replace(".yaml", ""),
        "AudioToMelSpectrogramPreprocessor": dict(_PRE),
        "JasperEncoder": {"activation": "relu", "conv_mask": True, "jasper": body},
        "labels": list(labels),
    }


def normalize_definition(d):
    """Accept both section spellings; drop keys the mel preprocessor never had (feat_type)."""
    d = copy.deepcopy(d)
    if "AudioToMelSpectrogramPreprocessor" not in d and "AudioPreprocessing" in d:
        pre = dict(d.pop("AudioPreprocessing"))
        pre.pop("feat_type", None)
        pre.setdefault("sample_rate", d.get("sample_rate", 16000))
        d["AudioToMelSpectrogramPreprocessor"] = pre
    return d


def load_model_definition(path):
    with open(path, encoding="utf-8") as f:
        return normalize_definition(yaml.safe_load(f))
