python3 profiles/diag_steps3.py 2>&1 | grep -v amdgpu.ids
