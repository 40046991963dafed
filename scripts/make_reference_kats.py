#!/usr/bin/env python3
"""Transcribes the reference's own known-answer DATA (inputs + expected
outputs only, no source text) from /root/reference/spec into
tests/golden/reference_kats.json and tests/golden/cedar_words.txt.

Run in the build container only (the GPU box has no /root/reference); the
outputs are committed.  The expected values are the literals the specs assert
(spec/ac_spec.cr:5-54, spec/ac_longest_match_spec.cr:5-63,
spec/cedar_spec.cr:4-13,28-293, README.md:29-38).
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def cedar_words():
    src = open(os.path.join(REF, "spec", "cedar_spec.cr"), encoding="utf-8").read()
    m = re.search(r"words = <<-TXT\n(.*?)\nTXT", src, re.S)
    return [l.strip() for l in m.group(1).split("\n")]


def main():
    words = cedar_words()
    assert len(words) == 249, len(words)
    with open(os.path.join(OUT, "cedar_words.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(words) + "\n")
    kats = {
        "_source": "literals asserted by the reference's specs; see scripts/make_reference_kats.py",
        "ac_match": [
            {"cite": "spec/ac_spec.cr:5-12, README.md:29-38", "keys": ["我", "我是", "是中"],
             "text": "我是中国人", "api": "string", "sep": None,
             "expect_end_value": [[1, 0], [2, 1], [3, 2]]},
            {"cite": "spec/ac_spec.cr:14-23 (match on the original object after save/load)",
             "keys": ["我", "我是", "是中"], "text": "我是中国人", "api": "string", "sep": None,
             "expect_end_value": [[1, 0], [2, 1], [3, 2]]},
            {"cite": "spec/ac_spec.cr:25-34", "keys": ["a", "aa"], "text": "a aaa", "api": "string",
             "sep": {"size": 256, "set": [32]}, "expect_end_value": [[1, 0]]},
            {"cite": "spec/ac_spec.cr:36-43", "keys": ["我", "我是", "是中"], "text": "我是中国人",
             "api": "chars", "sep": None, "expect_end_value": [[1, 0], [2, 1], [3, 2]]},
            {"cite": "spec/ac_spec.cr:45-54", "keys": ["a", "aa"], "text": "a aaa", "api": "chars",
             "sep": {"size": 256, "set": [32]}, "expect_end_value": [[1, 0]]},
        ],
        "ac_match_longest": [
            {"cite": "spec/ac_longest_match_spec.cr:5-18", "keys": ["Ruby", "ruby", "rub"],
             "text": "Ruby on rub", "intersectable": False,
             "expect": [[0, 4, "Ruby"], [8, 11, "rub"]]},
            {"cite": "spec/ac_longest_match_spec.cr:20-33", "keys": ["Ruby", "ruby", "uby "],
             "text": "ruby ", "intersectable": True,
             "expect": [[0, 4, "ruby"], [1, 5, "uby "]]},
        ],
        "cedar_insert_delete": {
            "cite": "spec/cedar_spec.cr:4-13",
            "ops": [["insert", "Ruby", 0], ["insert", "ruby", 1], ["insert", "rb", 2],
                    ["delete", "ruby", 1], ["delete", "ruby", -1], ["insert", "ruby", 1]],
        },
        "cedar_words": {"cite": "spec/cedar_spec.cr:28-293", "file": "cedar_words.txt", "count": 249},
    }
    with open(os.path.join(OUT, "reference_kats.json"), "w", encoding="utf-8") as f:
        json.dump(kats, f, ensure_ascii=False, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
