"""`freerec.data`: tags, fields, datasets and the chained datapipe surface."""
from . import datasets, fields, postprocessing, tags  # noqa: F401
