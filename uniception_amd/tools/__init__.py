"""Command-line helpers that ship with the package (checkpoint conversion)."""
