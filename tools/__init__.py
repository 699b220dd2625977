"""bench / profiling / golden-generation tooling (not product code)."""
