"""Offline stand-in for the `json5` package (absent in this image) -- enough for the reference's
config files: // comments, trailing commas, bare keys.  TEST INFRASTRUCTURE for running the
unmodified reference CLI; not part of the product."""
import json
import re


def _strip(text):
    out, i, n, in_str, q = [], 0, len(text), False, ""
    while i < n:
        c = text[i]
        if in_str:
            out.append(c)
            if c == "\\" and i + 1 < n:
                out.append(text[i + 1])
                i += 1
            elif c == q:
                in_str = False
        elif c in "\"'":
            in_str, q = True, c
            out.append(c)
        elif c == "/" and i + 1 < n and text[i + 1] == "/":
            while i < n and text[i] != "\n":
                i += 1
            continue
        elif c == "/" and i + 1 < n and text[i + 1] == "*":
            i = text.find("*/", i + 2)
            i = n if i < 0 else i + 2
            continue
        else:
            out.append(c)
        i += 1
    s = "".join(out)
    s = re.sub(r"([{,]\s*)([A-Za-z_][A-Za-z0-9_]*)(\s*:)", r'\1"\2"\3', s)  # bare keys
    s = re.sub(r",(\s*[}\]])", r"\1", s)  # trailing commas
    return s


def loads(text, **kw):
    return json.loads(_strip(text))


def load(fp, **kw):
    return loads(fp.read())


def dumps(obj, **kw):
    kw.pop("quote_keys", None)
    kw.pop("trailing_commas", None)
    return json.dumps(obj, **kw)


def dump(obj, fp, **kw):
    fp.write(dumps(obj, **kw))
