"""Trainer modules: each exposes `main()` and reads its settings from the environment
(the reference's contract, orchestration/orchestrator.py:286-291)."""
