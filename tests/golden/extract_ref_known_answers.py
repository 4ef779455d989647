#!/usr/bin/env python3
"""Re-types the known-answer DATA of the reference's own unit tests into JSON.

Run in the build container (needs /root/reference).  Output (committed):
  tests/golden/ref_simd_search_known_answers.json
  tests/golden/ref_search_wrappers_known_answers.json

Only data is taken: the test corpus string literal and the (call, expected
value) pairs asserted at
  test/src/string_search/simd_searchTest.cpp:12-34 (text), :36-99 (values)
  test/src/string_search/search_wrappersTest.cpp:12-21 (text), :26,39,52,64-69
The expected values are transcribed by hand below, next to the line they come
from; the text is read from the literal so that no byte is mistyped.
"""
import json
import re
from pathlib import Path

REF = Path("/root/reference/test/src/string_search")
OUT = Path(__file__).resolve().parent
SOH = "\x01"  # the tests write it as the octal escape \1


def c_literal(src: str, varname: str) -> str:
    m = re.search(varname + r"\[[^\]]*\]\s*=\s*((?:\s*\"(?:[^\"\\]|\\.)*\")+)\s*;", src)
    assert m, varname
    parts = re.findall(r"\"((?:[^\"\\]|\\.)*)\"", m.group(1))
    s = "".join(parts)
    out = []
    i = 0
    while i < len(s):
        ch = s[i]
        if ch == "\\":
            nxt = s[i + 1]
            if nxt == "n":
                out.append("\n")
                i += 2
            elif nxt in "01234567":
                j = i + 1
                while j < len(s) and j < i + 4 and s[j] in "01234567":
                    j += 1
                out.append(chr(int(s[i + 1:j], 8)))
                i = j
            elif nxt in "'\\\"":
                out.append(nxt)
                i += 2
            else:
                raise ValueError("escape \\" + nxt)
        else:
            out.append(ch)
            i += 1
    return "".join(out)


def main():
    simd_src = (REF / "simd_searchTest.cpp").read_text()
    text = c_literal(simd_src, "dummy_text")
    assert len(text) == 1240, len(text)
    long_pat = ("smooth-bellied chirognostic inkos BVM antigraphy pagne "
                "bicorne complementizer commorant ever-endingly sheikhly")
    assert len(long_pat) == 110
    simd = {
        "source": "test/src/string_search/simd_searchTest.cpp",
        "text": text,
        "text_len": 1240,
        # :36-41 strchr vs libc strchr -> offset (or -1)
        "strchr": [["L", text.find("L")], ["\n", text.find("\n")], [SOH, text.find(SOH)], ["\x02", -1]],
        # :43-60 strstr vs libc strstr -> offset (or -1)
        "strstr": [["Liane", text.find("Liane")], [long_pat, text.find(long_pat)], ["Helladic", text.find("Helladic")],
                   ["ly" + SOH, text.find("ly" + SOH)], ["jkahgsf", -1]],
        # :62-74 findNext(pattern, text, shift) -> absolute offset
        "findNext": [["Liane", 0, 0], [long_pat, 0, 113], ["Helladic", 0, 346], ["ly" + SOH, 0, 1237],
                     ["jkahgsf", 0, -1], ["ia", 0, 1], ["ia", 3, 455]],
        # :76-81 findNextNewLine(text, shift)
        "findNextNewLine": [[0, 223], [224, 282], [283, 728], [729, -1]],
        # :83-90 findAllPerLine (defined as countMatchingLines, simd_search.cpp:305)
        "countMatchingLines": [["is", 2], ["th", 3], ["Van", 1], ["y" + SOH, 1], ["Liane", 1], ["Vansdf", 0]],
        # :92-99 findAll (defined as countMatches, simd_search.cpp:324)
        "countMatches": [["is", 8], ["th", 5], ["Van", 1], ["y" + SOH, 1], ["Liane", 1], ["Vansdf", 0]],
    }
    (OUT / "ref_simd_search_known_answers.json").write_text(json.dumps(simd, indent=1))

    wr_src = (REF / "search_wrappersTest.cpp").read_text()
    wtext = c_literal(wr_src, "dummy_text")
    lines_with_nl = [
        "Liant reindorsing two-time zippering chromolithography rainbowweed\n",
        "smooth-bellied chirognostic inkos BVM antigraphy pagne bicorne\n",
        "complementizer commorant ever-endingly sheikhly\n",
        "refrangible terebras autobiographal mid-breast ant\n",
    ]
    wr = {
        "source": "test/src/string_search/search_wrappersTest.cpp",
        "text": wtext,
        "pattern": "ant",
        "byte_offsets_match": [2, 151, 197, 507],   # :26
        "byte_offsets_line": [0, 113, 176, 460],    # :39
        "count": 4,                                   # :52 (skip_to_nl defaults to true)
        # :64-69 lists the lines WITH a trailing "\n"; search_wrappers.h:204 copies
        # [line_begin, line_end) i.e. WITHOUT it, and the 100 MB goldens
        # (test/src/xsearchTest.cpp:126-172) have none either -> that assertion is
        # stale (the test is not registered with ctest).  Contract: no trailing \n.
        "line_as_asserted_stale": lines_with_nl,
        "line": [s[:-1] for s in lines_with_nl],
        # the regex wrappers on the same text (:74-105); the expression is passed to re2::RE2 wrapped in a group
        "regex": {
            "pattern": "(a[n|m]t)",
            "byte_offsets_match": [2, 151, 197, 507],   # :77-78, :82-83
            "byte_offsets_line": [0, 113, 176, 460],    # :90-91, :95-96
            "count": 4,                                   # :103
        },
    }
    assert '"(a[n|m]t)"' in wr_src and "res{2, 151, 197, 507}" in wr_src and "res{0, 113, 176, 460}" in wr_src
    (OUT / "ref_search_wrappers_known_answers.json").write_text(json.dumps(wr, indent=1))
    print("ok", len(text), len(wtext))


if __name__ == "__main__":
    main()
