"""Text normalisation used before WER/CER (same `VOCAB_SPECS` / `normalize_text` interface as the reference's
eval/utils.py:7-111).  The character classes are data: v0 = lower-case a-z, digits, umlauts, space (Swiss-German
ASR convention); v1-v3 keep case and add punctuation."""
from __future__ import annotations

import re
import string
from typing import Dict, Set

_SPACES = re.compile(r"[ \t]+")

_FOLD_BASE = dict(zip("áàâçéèêíìîñóòôúùûșş", "aaaceeeiiinooouuuss"))
_LOOKUP_V0 = {**_FOLD_BASE, "ß": "ss", "-": " ", "–": " ", "/": " "}
_LOOKUP_V1 = {**_LOOKUP_V0, **{k.upper(): v.upper() for k, v in _LOOKUP_V0.items()}}
_LOOKUP_V3 = {**{k: v for k, v in _FOLD_BASE.items() if k != "ş"}, "ß": "ss", "–": "-", "\xad": "-"}

_LOWER = string.ascii_lowercase + string.digits
_BOTH = string.ascii_lowercase + string.ascii_uppercase + string.digits

VOCAB_SPECS = {
    "v0": {"char_vocab": set(_LOWER + "äöü "), "char_lookup": _LOOKUP_V0, "transform_lowercase": True},
    "v1": {"char_vocab": set(_BOTH + "äöüÄÖÜ" + " .,:"), "char_lookup": _LOOKUP_V1, "transform_lowercase": False},
    "v2": {"char_vocab": set(_LOWER + "äöü" + " .,:"), "char_lookup": _LOOKUP_V1, "transform_lowercase": False},
    "v3": {"char_vocab": set(_BOTH + "äöüÄÖÜ" + " .,:-?!;"), "char_lookup": _LOOKUP_V3, "transform_lowercase": False},
}


def normalize_text(text: str, char_vocab: Set[str], char_lookup: Dict[str, str], transform_lowercase: bool = True) -> str:
    """lower-case (optional) -> fold characters via the lookup -> collapse blanks -> drop out-of-vocabulary
    characters -> collapse blanks again -> strip."""
    if transform_lowercase:
        text = text.lower()
    for src, dst in char_lookup.items():
        text = text.replace(src, dst)
    text = _SPACES.sub(" ", text)
    text = "".join(ch for ch in text if ch in char_vocab)
    return _SPACES.sub(" ", text).strip()
