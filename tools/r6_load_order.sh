cd $GRAFT_REPO_ROOT
echo "A: lib first, no torch";   python -c "
from secp256k1_zkp_amd import _native; _native.load()
from secp256k1_zkp_amd import Engine; e = Engine(0); print('ok A')" 2>&1 | tail -1
echo "B: lib first, then torch, then engine"; python -c "
from secp256k1_zkp_amd import _native; _native.load()
import torch; print(torch.cuda.is_available())
from secp256k1_zkp_amd import Engine; e = Engine(0); print('ok B')" 2>&1 | tail -2
echo "C: torch first"; python -c "
import torch
from secp256k1_zkp_amd import _native; _native.load()
from secp256k1_zkp_amd import Engine; e = Engine(0); print('ok C', torch.cuda.is_available())" 2>&1 | tail -1
echo "D: lib first + engine first, then torch"; python -c "
from secp256k1_zkp_amd import Engine; e = Engine(0); print('engine ok')
import torch; print('torch sees', torch.cuda.is_available(), torch.cuda.device_count())" 2>&1 | tail -2
ldd secp256k1_zkp_amd/libsecp256k1_zkp_amd.so | grep -i hip; python -c "import torch, os; print(os.path.dirname(torch.__file__))"; ls $(python -c "import torch, os; print(os.path.dirname(torch.__file__))")/lib | grep -i "amdhip\|hsa-runtime" | head
