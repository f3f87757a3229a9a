"""Stand-in for the third-party `ftfy` package (absent from this image; only used by the prompt cleaner, which the goldens bypass)."""


def fix_text(text, *a, **k):
    return text
