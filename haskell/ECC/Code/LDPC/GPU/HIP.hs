-- ECC.Code.LDPC.GPU.HIP -- binding of libldpc_hip.so (include/ldpc_hip.h) into the ku-fpg/ecc-ldpc plug-in
-- record, written against mkLDPC_CodeIO (src/ECC/Code/LDPC/Utils.hs:91-108) the way the CUDA plug-ins are
-- (src/ECC/Code/LDPC/GPU/CUDA/Arraylet2.hs:60-61).  Delivered as source: GHC is not available in the build
-- image, so this module has NOT been compiled.  See INTEGRATION.md for the cabal / Main.hs changes.
--
-- Codes exported (names resolve through the reference's grammar ldpc/<name>/<matrix>/<max-rounds>[/x/y]):
--   hip-tanh, hip-minsum                    QuasiCyclic Integer H  (.q files; what Fast.Arraylet takes, Fast/Arraylet.hs:68-79)
--   hip-tanh-bool, hip-minsum-bool          Matrix Bool H          (.alist / .m files; what Reference.Orig takes, Orig.hs:20-31)
--   hip-minsum-layered                      QuasiCyclic, row-layered schedule (an extension of this library)
--   hip-tanh-cuda32                         QuasiCyclic, the arithmetic of cuda-arraylet2 itself (float state, double product, float
--                                           atanh_ and clamp: LDPC_TANH_CUDA32, a parity mode on the flood path; r04)
-- Every code is built with maxThreadCount = hipThreads replicas (Utils.hs:53): replica i lives on GPU (i mod #GPUs), so
-- one Haskell process drives every GPU of the node; and the replicas of one GPU share a coalescing batcher
-- (ldpc_batcher_*): frames that several Haskell threads decode at the same moment go to the device in ONE launch.
{-# LANGUAGE ForeignFunctionInterface #-}
module ECC.Code.LDPC.GPU.HIP (codeTanh, codeMinSum, codeTanhBool, codeMinSumBool, codeMinSumLayered, codeMinSumF16PK,
                              codeTanhArrayletF64, codeTanhSparseF64, codeTanhCuda32, hipThreads) where

import ECC.Code.LDPC.Utils            (mkLDPC_CodeIO)
import ECC.Types
import qualified ECC.Code.LDPC.Fast.Encoder   as E
import qualified ECC.Code.LDPC.Reference.Orig as O
import qualified Data.Matrix.QuasiCyclic as Q
import qualified Data.Matrix as M
import qualified Data.Vector.Unboxed as U
import qualified Data.Vector.Storable as S
import qualified Data.Vector.Storable.Mutable as SM
import Data.Bits (testBit, popCount, shiftR)
import Data.Int  (Int32)
import Data.IORef
import Data.Word (Word8)
import Foreign.C.Types
import Foreign.C.String (CString, peekCString)
import Foreign.Ptr
import Foreign.Marshal.Alloc (alloca, allocaBytes)
import Foreign.Storable (peek, pokeByteOff)

data LdpcCode
data LdpcCtx
data LdpcBatcher

-- blocking calls (they wait for the GPU): `safe`, so other Haskell threads keep running meanwhile
foreign import ccall safe   "ldpc_init"               c_init       :: CInt -> IO CInt
foreign import ccall safe   "ldpc_shutdown"           c_shutdown   :: IO CInt
foreign import ccall unsafe "ldpc_device_count"       c_devCount   :: IO CInt
foreign import ccall unsafe "ldpc_last_error"         c_lastError  :: IO CString
foreign import ccall safe   "ldpc_code_create_qc"     c_codeQC     :: CInt -> CInt -> CInt -> Ptr Int32 -> IO (Ptr LdpcCode)
foreign import ccall safe   "ldpc_code_create_csr"    c_codeCSR    :: CInt -> CInt -> Ptr Int32 -> Ptr Int32 -> IO (Ptr LdpcCode)
foreign import ccall safe   "ldpc_ctx_create_cfg"     c_ctxCfg     :: Ptr LdpcCode -> Ptr () -> IO (Ptr LdpcCtx)
foreign import ccall safe   "ldpc_batcher_create"     c_batcher    :: Ptr LdpcCtx -> CInt -> CInt -> IO (Ptr LdpcBatcher)
foreign import ccall safe   "ldpc_batcher_decode_one" c_decodeOne  :: Ptr LdpcBatcher -> CInt -> Ptr Double -> Ptr Word8
                                                                   -> Ptr CInt -> Ptr CInt -> IO CInt

tanhRule, minSumRule, tanhCuda32Rule, f32, f64, f16pk, flooding, layered, pathAuto, sumReference, sumArraylet, sumSparse :: CInt
tanhCuda32Rule = 3   -- LDPC_TANH_CUDA32
tanhRule = 0; minSumRule = 1; f32 = 0; f64 = 1; f16pk = 3; flooding = 0; layered = 1; pathAuto = 0
sumReference = 0; sumArraylet = 1; sumSparse = 2      -- ldpc_sum_order: whose column-sum order the f64 parity modes follow

-- | maxThreadCount of these codes (Utils.hs:53): how many Haskell threads may decode at once.  The harness picks the
-- closure by ThreadId `rem` maxThreadCount (Utils.hs:63-69).
-- Measured through this call sequence (ecc_ldpc_amd/csrc/batcher.cc, jpl.4096, 50 turns, 16 host cores, callers = batch limit):
-- 64 threads 338-376 Mbit/s, 128: 408, 256: 467, 512: 339 -- the optimum; the host side sustains ~100 000 calls/s.
hipThreads :: Int
hipThreads = 256

-- | frames the batcher of one GPU collects at most, and how long (microseconds) a lone caller waits for company
coalesceFrames, coalesceWaitUs :: CInt
coalesceFrames = 256         -- = hipThreads: a launch can carry every caller that may be waiting (one workgroup per frame: 256 CUs)
coalesceWaitUs = 200

-- vars of the Code (the CUDA plug-ins keep their CudaAllocations there, Arraylet2.hs:287-293): the number of GPUs and,
-- per (graph, rule, schedule, GPU), the batcher its replicas share
data Vars = Vars { nDev :: Int, nextReplica :: IORef Int, batchers :: IORef [((String, (CInt, CInt, CInt, CInt), Int), Ptr LdpcBatcher)] }

initialize :: IO Vars
initialize = do
  n <- c_devCount
  if n <= 0 then error "libldpc_hip: no HIP device visible (there is no CPU fallback)" else return ()
  mapM_ (\d -> do rc <- c_init (fromIntegral d)           -- checks every device is a gfx950 GPU
                  if rc /= 0 then c_lastError >>= peekCString >>= error else return ()) [0 .. fromIntegral n - 1 :: Int]
  Vars (fromIntegral n) <$> newIORef 0 <*> newIORef []

finalize :: Vars -> IO ()
finalize _ = c_shutdown >> return ()

-- (rule, dtype, schedule, column-sum order)
type Mode = (CInt, CInt, CInt, CInt)

codeTanh, codeMinSum, codeMinSumLayered, codeTanhBool, codeMinSumBool, codeMinSumF16PK, codeTanhArrayletF64, codeTanhSparseF64 :: Code
codeTanh          = mkLDPC_CodeIO "hip-tanh"           hipThreads E.encoder (decoderQC   (tanhRule,   f32, flooding, sumReference)) initialize finalize
codeMinSum        = mkLDPC_CodeIO "hip-minsum"         hipThreads E.encoder (decoderQC   (minSumRule, f32, flooding, sumReference)) initialize finalize
codeMinSumLayered = mkLDPC_CodeIO "hip-minsum-layered" hipThreads E.encoder (decoderQC   (minSumRule, f32, layered,  sumReference)) initialize finalize
codeTanhBool      = mkLDPC_CodeIO "hip-tanh-bool"      hipThreads O.encoder (decoderBool (tanhRule,   f32, flooding, sumReference)) initialize finalize
codeMinSumBool    = mkLDPC_CodeIO "hip-minsum-bool"    hipThreads O.encoder (decoderBool (minSumRule, f32, flooding, sumReference)) initialize finalize
-- arithmetic in packed fp16, two frames per lane (LDPC_F16PK: the shipped AR4JA matrices)
codeMinSumF16PK   = mkLDPC_CodeIO "hip-minsum-f16pk"   hipThreads E.encoder (decoderQC   (minSumRule, f16pk, flooding, sumReference)) initialize finalize
-- Double twins of two of the reference's own decoders, last ulp included: run them next to `arraylet` / `sparse` in one harness
codeTanhArrayletF64 = mkLDPC_CodeIO "hip-tanh-f64-arraylet"    hipThreads E.encoder (decoderQC   (tanhRule, f64, flooding, sumArraylet)) initialize finalize
codeTanhSparseF64   = mkLDPC_CodeIO "hip-tanh-bool-f64-sparse" hipThreads O.encoder (decoderBool (tanhRule, f64, flooding, sumSparse))   initialize finalize
-- the live CUDA plug-in's own arithmetic (cudabits/arraylet2.cu:43-83, common.h:82-88,151-178), for comparing like with like (r04)
codeTanhCuda32 :: Code
codeTanhCuda32      = mkLDPC_CodeIO "hip-tanh-cuda32" hipThreads E.encoder (decoderQC (tanhCuda32Rule, f32, flooding, sumReference)) initialize finalize

-- rotation of the single set bit, -1 for an empty block (Fast/Arraylet.hs:68-79)
offsetsOf :: Q.QuasiCyclic Integer -> [Int32]
offsetsOf (Q.QuasiCyclic _ qm) = map f (M.toList qm)
  where f 0 = -1
        f n | popCount n == 1 = g n
            | otherwise       = error "QuasiCyclic matrix has non-powers of two initial value"
        g x | x `testBit` 0 = 0
            | otherwise     = 1 + g (x `shiftR` 1)

type Closure = Rate -> Int -> U.Vector Double -> IO (Maybe (U.Vector Bool))

-- `decoder vars h` is called maxThreadCount times by mkLDPC (Utils.hs:53); call number i serves GPU (i mod #GPUs)
decoderQC :: Mode -> Vars -> Q.QuasiCyclic Integer -> IO Closure
decoderQC mode vars h@(Q.QuasiCyclic sz qm) = do
  let offs = S.fromList (offsetsOf h)
      key  = "qc" ++ show (sz, M.nrows qm, M.ncols qm, S.toList offs)
      mk   = S.unsafeWith offs $ \p -> c_codeQC (fromIntegral sz) (fromIntegral (M.nrows qm)) (fromIntegral (M.ncols qm)) p
  closureFor vars key mode mk (sz * M.ncols qm)

-- the `Matrix Bool` decoders' input (Reference/Orig.hs:30-31): H as CSR, columns ascending inside a row
decoderBool :: Mode -> Vars -> M.Matrix Bool -> IO Closure
decoderBool mode vars h = do
  let rows   = [ [ fromIntegral (c - 1) :: Int32 | c <- [1 .. M.ncols h], h M.! (r, c) ] | r <- [1 .. M.nrows h] ]
      rowPtr = S.fromList (scanl (+) 0 (map (fromIntegral . length) rows)) :: S.Vector Int32
      colIdx = S.fromList (concat rows) :: S.Vector Int32
      key    = "csr" ++ show (M.nrows h, M.ncols h, S.toList colIdx)
      mk     = S.unsafeWith rowPtr $ \pr -> S.unsafeWith colIdx $ \pc ->
                 c_codeCSR (fromIntegral (M.nrows h)) (fromIntegral (M.ncols h)) pr pc
  closureFor vars key mode mk (M.ncols h)

-- one batcher (one ldpc_ctx behind it) per (graph, mode, GPU); every replica of that GPU shares it
closureFor :: Vars -> String -> Mode -> IO (Ptr LdpcCode) -> Int -> IO Closure
closureFor vars key mode@(rule, dtype, sched, sumOrder) mkCode n = do
  i <- atomicModifyIORef' (nextReplica vars) (\k -> (k + 1, k))
  let dev = i `mod` nDev vars
      k4  = (key, mode, dev)
  known <- lookup k4 <$> readIORef (batchers vars)
  b <- case known of
         Just b  -> return b
         Nothing -> do
           code <- mkCode
           if code == nullPtr then c_lastError >>= peekCString >>= error else return ()
           -- ldpc_ctx_config { size_t struct_size; int device, variant, dtype, max_batch, path, schedule, sum_order; } (40 bytes)
           ctx <- allocaBytes 40 $ \cfg -> do
                    pokeByteOff cfg 0  (40 :: CSize)
                    pokeByteOff cfg 8  (fromIntegral dev :: CInt)
                    pokeByteOff cfg 12 rule
                    pokeByteOff cfg 16 dtype
                    pokeByteOff cfg 20 coalesceFrames
                    pokeByteOff cfg 24 pathAuto
                    pokeByteOff cfg 28 sched
                    pokeByteOff cfg 32 sumOrder
                    pokeByteOff cfg 36 (0 :: CInt)
                    c_ctxCfg code cfg
           if ctx == nullPtr then c_lastError >>= peekCString >>= error else return ()
           b <- c_batcher ctx coalesceFrames coalesceWaitUs
           if b == nullPtr then c_lastError >>= peekCString >>= error else return ()
           atomicModifyIORef' (batchers vars) (\l -> ((k4, b) : l, ()))
           return b
  return $ \_rate maxI origLam -> do            -- origLam is already un-punctured (Utils.hs:55,69)
    let llr = S.convert origLam :: S.Vector Double
    bits <- SM.new n
    rc <- S.unsafeWith llr $ \pl -> SM.unsafeWith bits $ \pb ->
            alloca $ \pit -> alloca $ \pcv -> c_decodeOne b (fromIntegral maxI) pl pb pit pcv
    if rc /= 0
      then return Nothing                       -- harness substitutes hard(inp) (Utils.hs:70-71)
      else do out <- S.freeze bits
              return $ Just (U.map (/= 0) (S.convert out))
