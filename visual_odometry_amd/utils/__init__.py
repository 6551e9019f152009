"""Mirror of the reference's ``src/utils`` package for the modules on the hot path."""
