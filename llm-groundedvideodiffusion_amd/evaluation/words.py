"""The three inflections the benchmark prompts need (the reference uses the `inflect` package, which is not a dependency
here): cardinal words up to twenty, regular plurals, and the indefinite article.  Covers the benchmark vocabulary
(/root/reference/utils/eval/lvd.py:15-31,70-83); irregular nouns are out of scope by construction."""
import re

_CARDINALS = ("zero one two three four five six seven eight nine ten eleven twelve thirteen fourteen fifteen sixteen "
              "seventeen eighteen nineteen twenty").split()


def number_word(n):
    return _CARDINALS[n] if 0 <= n <= 20 else str(n)


def pluralize(phrase):
    """Plural of the head (last) noun of `phrase`."""
    if re.search(r"(s|x|z|ch|sh)$", phrase):
        return phrase + "es"
    if re.search(r"[^aeiou]y$", phrase):
        return phrase[:-1] + "ies"
    return phrase + "s"


def with_article(phrase):
    return ("an " if phrase[:1].lower() in "aeiou" else "a ") + phrase
