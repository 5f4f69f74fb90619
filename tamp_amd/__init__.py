"""tamp_amd placeholder (filled in below)."""
