import numpy as np

def schedule(N):
    s = {64:[16,4],128:[16,8],256:[16,16],512:[16,16,2],1024:[16,16,4],2048:[16,16,8],4096:[16,16,16],8192:[16,16,16,2]}
    return s[N]

def wg_fft(x, inverse=False):
    """Emulate: T=N/16 threads, reg[j][m] = x[j + m*T]; per stage scatter to 'LDS' then gather."""
    N = len(x); E = 16; T = N // E
    radices = schedule(N)
    sign = +1 if inverse else -1
    reg = np.array([[x[j + m*T] for m in range(E)] for j in range(T)], dtype=np.complex128)
    Ns = 1
    for si, R in enumerate(radices):
        B = E // R
        lds = np.zeros(N, dtype=np.complex128)
        out = np.zeros_like(reg)
        for j in range(T):
            for b in range(B):
                jv = j + b*T
                k = jv % Ns
                v = np.array([reg[j][b + r*B] for r in range(R)])
                tw = np.exp(sign*2j*np.pi*np.arange(R)*k/(Ns*R))
                v = v*tw
                V = np.array([np.sum(v*np.exp(sign*2j*np.pi*np.arange(R)*p/R)) for p in range(R)])
                base = (jv // Ns)*Ns*R + k
                for r in range(R):
                    lds[base + r*Ns] = V[r]
                    out[j][b + r*B] = V[r]
        Ns *= R
        if si == len(radices)-1:
            # last stage: registers already in load layout
            pos_ok = all(abs(lds[j+m*T]-out[j][m])<1e-9 for j in range(T) for m in range(E))
            assert pos_ok
            reg = out
        else:
            reg = np.array([[lds[j + m*T] for m in range(E)] for j in range(T)])
    X = np.zeros(N, dtype=np.complex128)
    for j in range(T):
        for m in range(E):
            X[j+m*T] = reg[j][m]
    return X

rng = np.random.default_rng(0)
for N in (64,128,256,512,1024,2048,4096,8192):
    x = rng.normal(size=N)+1j*rng.normal(size=N)
    X = wg_fft(x); Xi = wg_fft(x, True)
    print(N, np.abs(X-np.fft.fft(x)).max(), np.abs(Xi-np.fft.ifft(x)*N).max())

# centred transform sign trick
N=64
x = rng.normal(size=N)+1j*rng.normal(size=N)
ref = np.fft.fftshift(np.fft.fft(np.fft.fftshift(x), norm="ortho"))
s = (-1.0)**np.arange(N)
mine = s*np.fft.fft(s*x)/np.sqrt(N)
print("centred fwd", np.abs(ref-mine).max())
refi = np.fft.ifftshift(np.fft.ifft(np.fft.ifftshift(x), norm="ortho"))
minei = s*np.fft.ifft(s*x)*N/np.sqrt(N)
print("centred inv", np.abs(refi-minei).max())
