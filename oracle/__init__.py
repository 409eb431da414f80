"""TEST INFRASTRUCTURE ONLY.  CPU oracle (C restatement of the reference's trace loop) and the
tools that pin it to the reference.  Nothing in `ray_tracer_challenge_amd/` may import this."""
