from .dafne import DAFNe, DAFNeHead  # noqa: F401
from .dafne_outputs import DAFNeOutputs  # noqa: F401
