"""Argument / shape validation of the product path.  Plain `assert` disappears under `python -O`; these checks are the
error surface SURVEY.md §8b-4 describes (ValueError / TypeError / NotImplementedError with a message) and must not."""


def require(cond, msg, exc=ValueError):
    if not cond:
        raise exc(msg() if callable(msg) else msg)
