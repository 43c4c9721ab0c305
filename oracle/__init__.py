"""CPU oracle for the call_mods forward pass — TEST INFRASTRUCTURE (see ds_oracle.c header)."""
