from .dafne_outputs import DAFNeOutputs  # noqa: F401
