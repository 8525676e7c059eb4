"""Whisper's language table — `whisper.tokenizer.LANGUAGES` / `TO_LANGUAGE_CODE` (openai-whisper, not vendored by the
reference; imported at data/utils.py:11 and used by `_normalize_language_value`, data/utils.py:355-368).  The codes are the
`<|xx|>` tokens of `whisper_v3_utils/tokenizer.json` (100 languages, in token order)."""
from __future__ import annotations

_TABLE = (
    "en english, zh chinese, de german, es spanish, ru russian, ko korean, fr french, ja japanese, pt portuguese, tr turkish, "
    "pl polish, ca catalan, nl dutch, ar arabic, sv swedish, it italian, id indonesian, hi hindi, fi finnish, vi vietnamese, "
    "he hebrew, uk ukrainian, el greek, ms malay, cs czech, ro romanian, da danish, hu hungarian, ta tamil, no norwegian, "
    "th thai, ur urdu, hr croatian, bg bulgarian, lt lithuanian, la latin, mi maori, ml malayalam, cy welsh, sk slovak, "
    "te telugu, fa persian, lv latvian, bn bengali, sr serbian, az azerbaijani, sl slovenian, kn kannada, et estonian, "
    "mk macedonian, br breton, eu basque, is icelandic, hy armenian, ne nepali, mn mongolian, bs bosnian, kk kazakh, "
    "sq albanian, sw swahili, gl galician, mr marathi, pa punjabi, si sinhala, km khmer, sn shona, yo yoruba, so somali, "
    "af afrikaans, oc occitan, ka georgian, be belarusian, tg tajik, sd sindhi, gu gujarati, am amharic, yi yiddish, lo lao, "
    "uz uzbek, fo faroese, ht haitian creole, ps pashto, tk turkmen, nn nynorsk, mt maltese, sa sanskrit, lb luxembourgish, "
    "my myanmar, bo tibetan, tl tagalog, mg malagasy, as assamese, tt tatar, haw hawaiian, ln lingala, ha hausa, ba bashkir, "
    "jw javanese, su sundanese, yue cantonese"
)
LANGUAGES = {code: name for code, name in (item.split(" ", 1) for item in _TABLE.split(", "))}
TO_LANGUAGE_CODE = {name: code for code, name in LANGUAGES.items()}
TO_LANGUAGE_CODE.update({"burmese": "my", "valencian": "ca", "flemish": "nl", "haitian": "ht", "letzeburgesch": "lb",
                         "pushto": "ps", "panjabi": "pa", "moldavian": "ro", "moldovan": "ro", "sinhalese": "si",
                         "castilian": "es", "mandarin": "zh"})
