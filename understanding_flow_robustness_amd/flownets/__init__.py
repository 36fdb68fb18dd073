"""Flow networks of the reference's `models/` registry, rebuilt on the gfx950 operators."""
