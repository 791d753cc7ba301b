"""`freerec.data.tags`: what a field IS (USER / ITEM / LABEL ...) and what ROLE a forked field plays (SEQUENCE / POSITIVE / ...)."""


class _Tag(str):
    def __repr__(self):
        return f"<{str(self)}>"


_NAMES = ("USER", "ITEM", "ID", "RATING", "TIMESTAMP", "LABEL", "FEATURE", "SPARSE", "DENSE", "EMBED", "SEQUENCE", "POSITIVE", "NEGATIVE",
          "UNSEEN", "SEEN", "SIZE", "MATCHING", "NEXTITEM", "PREDICTION")
for _n in _NAMES:
    globals()[_n] = _Tag(_n)
__all__ = list(_NAMES)
