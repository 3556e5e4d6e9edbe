"""Mirror of the reference's utils package (only the hot-path functions)."""
