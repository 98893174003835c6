"""Drop-in namespace: `import q1physrl_env.env` / `q1physrl_env.phys` resolve to the MI355X implementation,
so code written against the reference package (train.py:49-51, analyse.py, mkdemo.py) runs unchanged."""
