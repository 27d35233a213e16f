"""TEST INFRASTRUCTURE ONLY: CPU oracle for the ray-trace hot path (see trace_oracle.py)."""
