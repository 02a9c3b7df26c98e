-- ECC.Code.LDPC.GPU.HIP -- binding of libldpc_hip.so (include/ldpc_hip.h) into the ku-fpg/ecc-ldpc plug-in
-- record, written against mkLDPC_CodeIO (src/ECC/Code/LDPC/Utils.hs:91-108) the way the CUDA plug-ins are
-- (src/ECC/Code/LDPC/GPU/CUDA/Arraylet2.hs:60-61).  Delivered as source: GHC is not available in the build
-- image, so this module has NOT been compiled.  See INTEGRATION.md for the cabal / Main.hs changes.
{-# LANGUAGE ForeignFunctionInterface #-}
module ECC.Code.LDPC.GPU.HIP (codeTanh, codeMinSum) where

import ECC.Code.LDPC.Utils            (mkLDPC_CodeIO)
import ECC.Types
import qualified ECC.Code.LDPC.Fast.Encoder as E
import qualified Data.Matrix.QuasiCyclic as Q
import qualified Data.Matrix as M
import qualified Data.Vector.Unboxed as U
import qualified Data.Vector.Storable as S
import qualified Data.Vector.Storable.Mutable as SM
import Data.Bits (testBit, popCount, shiftR)
import Data.Int  (Int32)
import Data.Word (Word8)
import Foreign.C.Types
import Foreign.C.String (CString, peekCString)
import Foreign.Ptr
import Foreign.Marshal.Alloc (alloca)
import Foreign.Storable (peek)

data LdpcCode
data LdpcCtx

-- blocking calls (they wait for the GPU): `safe`
foreign import ccall safe   "ldpc_init"           c_init      :: CInt -> IO CInt
foreign import ccall safe   "ldpc_shutdown"       c_shutdown  :: IO CInt
foreign import ccall unsafe "ldpc_last_error"     c_lastError :: IO CString
foreign import ccall safe   "ldpc_code_create_qc" c_codeQC    :: CInt -> CInt -> CInt -> Ptr Int32 -> IO (Ptr LdpcCode)
foreign import ccall safe   "ldpc_ctx_create"     c_ctxCreate :: Ptr LdpcCode -> CInt -> CInt -> CInt -> IO (Ptr LdpcCtx)
foreign import ccall safe   "ldpc_decode_one"     c_decodeOne :: Ptr LdpcCtx -> CInt -> Ptr Double -> Ptr Word8
                                                              -> Ptr CInt -> Ptr CInt -> IO CInt

tanhRule, minSumRule, f32 :: CInt
tanhRule = 0; minSumRule = 1; f32 = 0

-- same shape as CUDAArraylet2.code (GPU/CUDA/Arraylet2.hs:60-61): one replica, Fast.Encoder
codeTanh, codeMinSum :: Code
codeTanh   = mkLDPC_CodeIO "hip-tanh"   1 E.encoder (decoder tanhRule)   initialize finalize
codeMinSum = mkLDPC_CodeIO "hip-minsum" 1 E.encoder (decoder minSumRule) initialize finalize

initialize :: IO ()
initialize = do rc <- c_init 0
                if rc /= 0 then c_lastError >>= peekCString >>= error else return ()

finalize :: () -> IO ()
finalize _ = c_shutdown >> return ()

-- rotation of the single set bit, -1 for an empty block (Fast/Arraylet.hs:68-79)
offsetsOf :: Q.QuasiCyclic Integer -> [Int32]
offsetsOf (Q.QuasiCyclic _ qm) = map f (M.toList qm)
  where f 0 = -1
        f n | popCount n == 1 = g n
            | otherwise       = error "QuasiCyclic matrix has non-powers of two initial value"
        g x | x `testBit` 0 = 0
            | otherwise     = 1 + g (x `shiftR` 1)

decoder :: CInt -> () -> Q.QuasiCyclic Integer
        -> IO (Rate -> Int -> U.Vector Double -> IO (Maybe (U.Vector Bool)))
decoder rule _ h@(Q.QuasiCyclic sz qm) = do
  let offs = S.fromList (offsetsOf h)
  code <- S.unsafeWith offs $ \p ->
            c_codeQC (fromIntegral sz) (fromIntegral (M.nrows qm)) (fromIntegral (M.ncols qm)) p
  ctx  <- c_ctxCreate code rule f32 1
  if ctx == nullPtr then c_lastError >>= peekCString >>= error else return ()
  let n = sz * M.ncols qm
  return $ \_rate maxI origLam -> do            -- origLam is already un-punctured (Utils.hs:55,69)
    let llr = S.convert origLam :: S.Vector Double
    bits <- SM.new n
    rc <- S.unsafeWith llr $ \pl -> SM.unsafeWith bits $ \pb ->
            alloca $ \pit -> alloca $ \pcv -> c_decodeOne ctx (fromIntegral maxI) pl pb pit pcv
    if rc /= 0
      then return Nothing                       -- harness substitutes hard(inp) (Utils.hs:70-71)
      else do out <- S.freeze bits
              return $ Just (U.map (/= 0) (S.convert out))
