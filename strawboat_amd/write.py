"""write:: — page encode API (filled in by sb_encode; see write_columns)."""
