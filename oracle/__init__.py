"""CPU oracle: test infrastructure only (see pc_oracle.c header)."""
