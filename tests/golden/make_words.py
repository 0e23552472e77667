#!/usr/bin/env python3
"""Write tests/golden/words.txt: the word-list fixture that stands in for /usr/share/dict/words, which the
reference's strgen reads (strgen.cc:34) and which is absent from the build image and the GPU box.
1200 distinct pseudo-words, one per line, in dictionary order like the real file (capitalised entries and
possessives first): enough for create_strvec(N) up to N = 1200^2.  Deterministic; data only."""
import os

ON = ["b", "br", "c", "ch", "cl", "d", "dr", "f", "fl", "g", "gr", "h", "j", "k", "l", "m", "n", "p", "pl", "pr",
      "qu", "r", "s", "sh", "sl", "st", "t", "th", "tr", "v", "w", "z", ""]
VO = ["a", "e", "i", "o", "u", "ai", "ea", "ou", "oo", "y"]
CO = ["b", "ck", "d", "ft", "g", "l", "ll", "m", "n", "nd", "ng", "p", "r", "rt", "s", "sh", "st", "t", "x", ""]


def word(i):
    x = (i * 2654435761 + 12345) & 0xFFFFFFFF
    parts = []
    for syll in range(2 + (x >> 29) % 2):
        parts.append(ON[(x >> (5 * syll)) % len(ON)] + VO[(x >> (5 * syll + 3)) % len(VO)] + CO[(x >> (5 * syll + 7)) % len(CO)])
    return "".join(parts)


words, i = set(), 0
while len(words) < 1200:
    w = word(i)
    i += 1
    if len(w) < 2:
        continue
    if i % 11 == 0:
        w = w.capitalize()
    if i % 17 == 0:
        w += "'s"
    words.add(w)
out = sorted(words)
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "words.txt"), "w") as f:
    f.write("\n".join(out) + "\n")
print(len(out), "words,", sum(len(w) + 1 for w in out), "bytes; first:", out[:5], "last:", out[-3:])
